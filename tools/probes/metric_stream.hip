// Probe (tools only): the block kernel's METRIC STREAM alone -- same persistent grid (3 workgroups of 256 threads per CU, contiguous ranges of 4x4x4 bricks),
// same lane shape (25 of 32 lanes per cell, two 16-byte loads and one 8-byte load per plane and lane: the pair layout of bp5_kernels.hpp), six planes,
// 8 cells per pass -- but no gather, no contractions, no write-out.  What rate does the pattern itself reach, and does the layout of the six planes matter?
//   layout 0  plane-major [plane][cell][125]  (the library's: six streams 1.6 GB apart, a pass = one 8 KB run per plane)
//   layout 1  pass-major  [pass][plane][8 cells][125]  (a pass = one 48 KB run)
//   layout 2  cell-major  [cell][plane][125]  (round 2's probe: 6 KB per cell)
//   layout 3  the same bytes as ONE linear stream of whole 1 KB wave-instructions (all 64 lanes, 16 bytes each): the ideal
// Build: hipcc --offload-arch=gfx950 -O3 -o metric_stream metric_stream.hip ; run: timeout -k 10 120 ./metric_stream
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2), aligned(8)));

template <int LAYOUT, int INFLIGHT>
__global__ void __launch_bounds__(256, 3) stream_kernel(const double *coef, size_t n_cells, size_t n_pass, double *sink)
{
  const int t = threadIdx.x, slot = t >> 5, ab = t & 31;
  const bool lane_ok = ab < 25;
  const size_t per = (n_pass + gridDim.x - 1) / gridDim.x;
  const size_t p0 = (size_t)blockIdx.x * per, p1 = std::min(n_pass, p0 + per);
  double acc = 0.0;
  const size_t plane = n_cells * 125;
  for (size_t ps = p0; ps < p1; ++ps) {
    const size_t cell = ps * 8 + slot;
    double S[6][5];
    if (LAYOUT == 3) {
      const v2d *base = reinterpret_cast<const v2d *>(coef) + ps * (6 * 8 * 125 / 2); // 3000 pairs per pass
#pragma unroll
      for (int k = 0; k < 12; ++k) { // 12 x 256 threads x 16 B = 48 KB (3072 pairs: 72 beyond the pass, inside the buffer)
        const v2d v = base[k * 256 + t];
        S[k / 2][(2 * k) % 4] = v.x; S[k / 2][(2 * k) % 4 + 1] = v.y;
      }
#pragma unroll
      for (int pl = 0; pl < 6; ++pl) S[pl][4] = 0.0;
    } else {
#pragma unroll
      for (int pl = 0; pl < 6; ++pl) {
        const double *cf = LAYOUT == 0 ? coef + pl * plane + cell * 125 : LAYOUT == 1 ? coef + (ps * 6 + pl) * 1000 + slot * 125 : coef + (cell * 6 + pl) * 125;
        if (lane_ok) {
          const v2d a = *reinterpret_cast<const v2d *>(cf + 2 * ab), b = *reinterpret_cast<const v2d *>(cf + 50 + 2 * ab);
          S[pl][0] = a.x; S[pl][1] = a.y; S[pl][2] = b.x; S[pl][3] = b.y; S[pl][4] = cf[100 + ab];
        } else {
#pragma unroll
          for (int i = 0; i < 5; ++i) S[pl][i] = 0.0;
        }
      }
    }
    if (INFLIGHT) __builtin_amdgcn_s_sleep(INFLIGHT); // (a stand-in for work between issue and use)
#pragma unroll
    for (int pl = 0; pl < 6; ++pl)
#pragma unroll
      for (int i = 0; i < 5; ++i) acc += S[pl][i];
  }
  if (acc == 1.2345e300) *sink = acc;
}

template <int LAYOUT>
static void run(const double *coef, size_t n_cells, double *sink, int n_wg)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t n_pass = n_cells / 8;
  std::vector<float> ms;
  for (int r = -2; r < 9; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((stream_kernel<LAYOUT, 0>), dim3(n_wg), dim3(256), 0, 0, coef, n_cells, n_pass, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (r >= 0) ms.push_back(t);
  }
  CK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  const double bytes = (double)n_cells * 6 * 125 * 8;
  static const char *nm[4] = {"plane-major (library)", "pass-major", "cell-major", "linear 1 KB instructions"};
  printf("layout %d %-26s grid %4d  median %.4f ms  best %.4f ms  %7.1f GB/s\n", LAYOUT, nm[LAYOUT], n_wg, ms[ms.size() / 2], ms[0], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
  fflush(stdout);
}

int main()
{
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const size_t n_cells = (size_t)116 * 116 * 116 / 8 * 8;
  double *coef, *sink;
  CK(hipMalloc((void **)&coef, n_cells * 6 * 125 * 8 + 4096));
  CK(hipMemset(coef, 0, n_cells * 6 * 125 * 8 + 4096));
  CK(hipMalloc((void **)&sink, 8));
  printf("%s, %d CUs; metric of %zu cells, p = 4: %.2f GB\n", prop.gcnArchName, prop.multiProcessorCount, n_cells, n_cells * 6e3 / 1e9);
  for (int k : {3, 2, 1}) {
    const int n_wg = k * prop.multiProcessorCount;
    run<0>(coef, n_cells, sink, n_wg);
    run<1>(coef, n_cells, sink, n_wg);
    run<2>(coef, n_cells, sink, n_wg);
    run<3>(coef, n_cells, sink, n_wg);
  }
  return 0;
}
