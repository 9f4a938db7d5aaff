"""Counters of the binned table-gradient backward's pass A (f2n_debug_bin_stats): how many tiles took
the combine route per level and what it turned their contributions into (tools only).

  stats = bin_stats.enable(lib, dev) ... run ... bin_stats.report(lib, stats, L)
"""
import ctypes
import sys

import torch


def enable(cdll, dev):
    stats = torch.zeros(64, 4, dtype=torch.int32, device=dev)
    cdll.f2n_debug_bin_stats.argtypes = [ctypes.c_void_p]
    cdll.f2n_debug_bin_stats.restype = None
    cdll.f2n_debug_bin_stats(stats.data_ptr())
    return stats


def report(cdll, stats, L, out=sys.stdout):
    torch.cuda.synchronize()
    cdll.f2n_debug_bin_stats(None)
    st = stats.cpu()
    if int(st[1, 3]):
        print("    %d tiles, %.0f lanes per tile repeat a level-0 cell of the four lanes before them" % (int(st[1, 3]), int(st[0, 3]) / int(st[1, 3])),
              file=out)
    for l in range(L):
        if not int(st[l, 0]) and (int(st[l, 2]) or int(st[l, 3])):
            print("    level %2d: split pass overflow: %d records past a queue, %d past a run" % (l, int(st[l, 3]), int(st[l, 2])),
                  file=out)
        if int(st[l, 0]):
            print("    level %2d: %6d tiles combined, %5.0f non-zero contributions -> %5.0f records per tile"
                  % (l, int(st[l, 0]), int(st[l, 1]) / int(st[l, 0]), int(st[l, 2]) / int(st[l, 0]))
                  + ("  (%.0f of them contributions that lost their slot of the table)" % (int(st[32 + l, 0]) / int(st[l, 0]))
                     if int(st[32 + l, 0]) else ""), file=out)
