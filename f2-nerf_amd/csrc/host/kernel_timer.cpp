// kernel_timer.cpp -- see kernel_timer.hpp.
#include "kernel_timer.hpp"

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <map>
#include <mutex>

namespace f2n
{

namespace
{

struct Record
{
  const char * name;
  hipEvent_t start, stop;
  double units;
};

std::atomic<bool> g_enabled{false};
std::mutex g_mu;  // forward runs on the caller thread, backward on the autograd device thread
std::vector<Record> g_records;

}  // namespace

void kernel_timer_enable(bool on) { g_enabled.store(on, std::memory_order_relaxed); }
bool kernel_timer_enabled() { return g_enabled.load(std::memory_order_relaxed); }

ScopedKernelTimer::ScopedKernelTimer(const char * name, void * stream, double units)
{
  if (!kernel_timer_enabled()) return;
  Record r{name, nullptr, nullptr, units};
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  hipEventRecord(r.start, (hipStream_t)stream);
  std::lock_guard<std::mutex> lk(g_mu);
  g_records.push_back(r);
  slot_ = (int)g_records.size() - 1;
  stream_ = stream;
}

ScopedKernelTimer::~ScopedKernelTimer()
{
  if (slot_ < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot_ < (int)g_records.size()) hipEventRecord(g_records[slot_].stop, (hipStream_t)stream_);
}

std::vector<KernelTiming> kernel_timer_collect()
{
  std::vector<Record> recs;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    recs.swap(g_records);
  }
  std::map<std::string, KernelTiming> acc;
  for (auto & r : recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.stop) == hipSuccess &&
        hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      auto & t = acc[r.name];
      t.name = r.name;
      t.launches += 1;
      t.total_ms += ms;
      t.units += r.units;
    }
    hipEventDestroy(r.start);
    hipEventDestroy(r.stop);
  }
  std::vector<KernelTiming> out;
  for (auto & kv : acc) out.push_back(kv.second);
  return out;
}

}  // namespace f2n
