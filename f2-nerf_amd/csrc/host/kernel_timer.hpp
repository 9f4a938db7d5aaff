// kernel_timer.hpp -- opt-in hipEvent timing of individual C-ABI kernel launches, on the stream they
// are launched on.  bench.py switches it on for the timed region to obtain the per-launch duration
// its roofline figure needs (torch.cuda.Event would only see torch's current stream; these events
// are recorded on exactly the stream handed to the kernel).  Disabled = one relaxed atomic load.
#pragma once

#include <string>
#include <vector>

namespace f2n
{

struct KernelTiming
{
  std::string name;
  int64_t launches = 0;
  double total_ms = 0.0;
  double units = 0.0;  // caller-defined work units summed over launches (samples, rays ...)
};

void kernel_timer_enable(bool on);
bool kernel_timer_enabled();
// Waits for the recorded events, folds them into per-name totals and clears the event list.
std::vector<KernelTiming> kernel_timer_collect();

// RAII: records a start event now and a stop event at scope exit, both on `stream`.
class ScopedKernelTimer
{
public:
  ScopedKernelTimer(const char * name, void * stream, double units);
  ~ScopedKernelTimer();

private:
  int slot_ = -1;
  void * stream_ = nullptr;
};

}  // namespace f2n
