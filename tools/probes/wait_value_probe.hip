// Probe (tools only, not part of the library): can a stream wait for a value that a RUNNING kernel on another stream writes,
// and how long after the write does the waiting stream's next kernel start?  Two mechanisms:
//   A  hipStreamWaitValue64 on signal memory (hipExtMallocWithFlags(hipMallocSignalMemory)) and on plain device memory
//   B  a one-wave spin kernel (bounded: gives up after max_spins) on the waiting stream
// Build: hipcc --offload-arch=gfx950 -O2 -o wait_value_probe wait_value_probe.hip ; run under `timeout -k 5 60`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(unsigned long long *flag, unsigned long long value, unsigned long long *stamps, long long busy_ticks, int n_wg_total)
{
  // every workgroup burns busy_ticks of the 100 MHz wall clock, then counts itself in; the counter reaches `value` when all are done
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < busy_ticks) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned long long old = atomicAdd(flag, 1ull);
    if (old + 1 == value) stamps[0] = wall_clock64(); // the moment the value is complete
    // keep the kernel alive a while longer: the consumer must start BEFORE this kernel ends to prove mid-kernel signalling
    const unsigned long long t1 = wall_clock64();
    while ((long long)(wall_clock64() - t1) < 4 * busy_ticks) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0) stamps[2] = wall_clock64(); // end of (this workgroup of) the producer
  }
}
__global__ void consumer(unsigned long long *stamps) { if (threadIdx.x == 0) stamps[1] = wall_clock64(); }
__global__ void spin_wait(const unsigned long long *flag, unsigned long long value, long long max_spins, int *timed_out)
{
  long long k = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
    if (++k > max_spins) { *timed_out = 1; return; }
    __builtin_amdgcn_s_sleep(16);
  }
}

int main()
{
  int can = -1;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t sa, sb;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
  unsigned long long *sig = nullptr, *plain = nullptr, *stamps = nullptr, h[3];
  int *to = nullptr;
  CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
  CK(hipMalloc((void **)&plain, 8));
  CK(hipMalloc((void **)&stamps, 64));
  CK(hipMalloc((void **)&to, 4));
  const int n_wg = 768;
  for (int mode = 0; mode < 3; ++mode) {
    if (mode < 2 && can != 1) continue;
    unsigned long long *flag = mode == 0 ? sig : plain;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(flag, 0, 8)); CK(hipMemset(stamps, 0, 64)); CK(hipMemset(to, 0, 4));
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(producer, dim3(n_wg), dim3(256), 0, sa, flag, (unsigned long long)n_wg, stamps, 5000LL /* 50 us */, n_wg);
      if (mode < 2) CK(hipStreamWaitValue64(sb, flag, (uint64_t)n_wg, hipStreamWaitValueGte, ~0ull));
      else hipLaunchKernelGGL(spin_wait, dim3(1), dim3(64), 0, sb, flag, (unsigned long long)n_wg, 2000000LL, to);
      hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, sb, stamps);
      CK(hipGetLastError());
      CK(hipDeviceSynchronize());
      int hto = 0;
      CK(hipMemcpy(h, stamps, 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hto, to, 4, hipMemcpyDeviceToHost));
      printf("mode %d (%s) rep %d: value complete -> consumer start %+.1f us; consumer start -> producer end %+.1f us%s\n", mode,
             mode == 0 ? "hipStreamWaitValue64, signal memory" : mode == 1 ? "hipStreamWaitValue64, plain device memory" : "spin kernel",
             rep, ((double)h[1] - (double)h[0]) / 100.0, ((double)h[2] - (double)h[1]) / 100.0, hto ? "  [spin TIMED OUT]" : "");
    }
  }
  return 0;
}
