// Probe (tools only, not part of the library): what streaming rate does THIS box reach, and with which kernel shape?
// the library's copy kernel (grid-stride, ONE 16-byte load in flight per lane, 2048-block cap) reads 4.9-5.2 TB/s where
// the microarchitecture guide quotes 6.29 TB/s for a float4 copy.  This sweep separates "the pool's boxes" from "the kernel shape":
//   ops    read (1R) | fill (1W) | copy (1R 1W) | triad (2R 1W) | upd1 (3R 2W = cgm_update_kernel<1>) | upd2 (4R 3W = cgm_update_kernel<2>)
//   U      independent 16-byte accesses per lane, stream and trip (1, 2, 4, 8): all loads of a trip are issued before its first store
//   nt     non-temporal loads/stores on or off
//   VB     256 / 512 threads per workgroup
//   grid   k x CUs workgroups walking the arrays block-cyclically (k = 1 ... 32), "flat" = one trip per workgroup, and "chunk" = k x CUs
//          workgroups each streaming ONE contiguous range
// Arrays: 3 * 2^24 double2 = 805 MB each (well past the 256 MB memory-side cache).  Timed with HIP events, REPS launches, median and best.
// Build: hipcc --offload-arch=gfx950 -O3 -o hbm_sweep hbm_sweep.hip ; run: timeout -k 10 300 ./hbm_sweep [quick]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
enum { OP_READ, OP_FILL, OP_COPY, OP_TRIAD, OP_UPD1, OP_UPD2, N_OPS };
static const char *op_name[N_OPS] = {"read", "fill", "copy", "triad", "upd1", "upd2"};
static const int op_streams[N_OPS] = {1, 1, 2, 3, 5, 7}; // 16-byte streams crossing HBM per element pair

template <bool NT> __device__ __forceinline__ v2d ld(const v2d *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(v2d *p, v2d v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// a, b, d writable, c read-only.  n2 = number of double2 per array, a multiple of VB * U * gridDim.x for MODE_CHUNK.
template <int OP, int U, bool NT, int VB>
__global__ void __launch_bounds__(VB) sweep_kernel(v2d *a, v2d *b, const v2d *c, v2d *d, size_t n2, int chunked, double s, double *sink)
{
  const size_t trips = n2 / ((size_t)VB * U);
  size_t t0 = blockIdx.x, t1 = trips, dt = gridDim.x;
  if (chunked) { const size_t per = trips / gridDim.x; t0 = blockIdx.x * per; t1 = t0 + per; dt = 1; }
  v2d acc = {0.0, 0.0};
  for (size_t t = t0; t < t1; t += dt) {
    const size_t base = t * VB * U + threadIdx.x;
    v2d av[U], bv[U], cv[U], dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * VB;
      if (OP == OP_READ) av[u] = ld<NT>(a + i);
      if (OP == OP_COPY || OP == OP_TRIAD) bv[u] = ld<NT>(b + i);
      if (OP == OP_TRIAD) cv[u] = ld<NT>(c + i);
      if (OP == OP_UPD1 || OP == OP_UPD2) { av[u] = ld<false>(a + i); bv[u] = ld<false>(b + i); cv[u] = ld<NT>(c + i); }
      if (OP == OP_UPD2) dv[u] = ld<NT>(d + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * VB;
      if (OP == OP_READ) acc += av[u];
      if (OP == OP_FILL) st<NT>(a + i, v2d{s, s});
      if (OP == OP_COPY) st<NT>(a + i, bv[u]);
      if (OP == OP_TRIAD) st<NT>(a + i, bv[u] + s * cv[u]);
      if (OP == OP_UPD1 || OP == OP_UPD2) { // r += alpha v ; p = beta p - r (; x += alpha p): p and r stay cacheable, v and x stream
        if (OP == OP_UPD2) st<NT>(d + i, dv[u] + s * av[u]);
        const v2d rn = bv[u] + s * cv[u];
        st<false>(b + i, rn);
        st<false>(a + i, s * av[u] - rn);
      }
    }
  }
  if (OP == OP_READ && acc.x + acc.y == 1.2345e300) *sink = acc.x;
}

struct Result { int op, U, nt, vb, k; const char *mode; double ms_med, ms_min, gbs; };

template <int OP, int U, bool NT, int VB>
static void run_one(std::vector<Result> &out, v2d *a, v2d *b, v2d *c, v2d *d, double *sink, size_t n2, int n_cus, int reps, bool quick)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t trips = n2 / ((size_t)VB * U);
  struct G { const char *mode; int k; size_t grid; int chunked; };
  std::vector<G> grids;
  for (int k : {1, 2, 4, 8, 16, 32}) if (!quick || k == 4 || k == 8) grids.push_back({"cyclic", k, (size_t)k * n_cus, 0});
  for (int k : {4, 8, 16, 64}) if (trips % ((size_t)k * n_cus) == 0 && (!quick || k == 8)) grids.push_back({"chunk", k, (size_t)k * n_cus, 1});
  grids.push_back({"flat", 0, trips, 0});
  for (const G &g : grids) {
    std::vector<float> ms(reps);
    for (int r = -2; r < reps; ++r) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL((sweep_kernel<OP, U, NT, VB>), dim3((unsigned)g.grid), dim3(VB), 0, 0, a, b, c, d, n2, g.chunked, 0.999, sink);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (r >= 0) ms[r] = t;
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)op_streams[OP] * 16.0 * n2;
    Result res{OP, U, NT, VB, g.k, g.mode, ms[reps / 2], ms[0], bytes / (ms[reps / 2] * 1e-3) / 1e9};
    out.push_back(res);
    printf("%-5s U=%d nt=%d VB=%4d %-6s k=%2d grid=%8zu  med %.4f ms  best %.4f ms  %7.1f GB/s (best %7.1f)\n", op_name[OP], U, (int)NT, VB, g.mode, g.k,
           g.grid, res.ms_med, res.ms_min, res.gbs, bytes / (ms[0] * 1e-3) / 1e9);
    fflush(stdout);
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int OP, int U>
static void run_u(std::vector<Result> &out, v2d *a, v2d *b, v2d *c, v2d *d, double *sink, size_t n2, int n_cus, int reps, bool quick)
{
  run_one<OP, U, false, 256>(out, a, b, c, d, sink, n2, n_cus, reps, quick);
  run_one<OP, U, true, 256>(out, a, b, c, d, sink, n2, n_cus, reps, quick);
  if (!quick) {
    run_one<OP, U, false, 512>(out, a, b, c, d, sink, n2, n_cus, reps, quick);
    run_one<OP, U, true, 512>(out, a, b, c, d, sink, n2, n_cus, reps, quick);
  }
}
template <int OP>
static void run_op(std::vector<Result> &out, v2d *a, v2d *b, v2d *c, v2d *d, double *sink, size_t n2, int n_cus, int reps, bool quick)
{
  run_u<OP, 1>(out, a, b, c, d, sink, n2, n_cus, reps, quick);
  run_u<OP, 2>(out, a, b, c, d, sink, n2, n_cus, reps, quick);
  run_u<OP, 4>(out, a, b, c, d, sink, n2, n_cus, reps, quick);
  if (OP <= OP_TRIAD) run_u<OP, 8>(out, a, b, c, d, sink, n2, n_cus, reps, quick); // 4 streams x 8 x 4 VGPRs would spill
}

int main(int argc, char **argv)
{
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d MHz, memory clock %d MHz, bus %d bit\n", prop.gcnArchName, n_cus, prop.clockRate / 1000, prop.memoryClockRate / 1000,
         prop.memoryBusWidth);
  const size_t n2 = (size_t)3 << 24;
  v2d *buf[4];
  double *sink;
  for (auto &p : buf) { CK(hipMalloc((void **)&p, n2 * sizeof(v2d))); CK(hipMemset(p, 0, n2 * sizeof(v2d))); }
  CK(hipMalloc((void **)&sink, 8));
  // hipMemcpy D2D of the runtime as a reference point
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(e0, 0)); CK(hipMemcpyAsync(buf[0], buf[1], n2 * sizeof(v2d), hipMemcpyDeviceToDevice, 0)); CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); best = std::min(best, t);
    }
    printf("hipMemcpyAsync D2D 805 MB: best %.4f ms = %.1f GB/s (read + write)\n", best, 2.0 * n2 * 16 / (best * 1e-3) / 1e9);
  }
  std::vector<Result> out;
  const int reps = quick ? 5 : 9;
  run_op<OP_READ>(out, buf[0], buf[1], buf[2], buf[3], sink, n2, n_cus, reps, quick);
  run_op<OP_FILL>(out, buf[0], buf[1], buf[2], buf[3], sink, n2, n_cus, reps, quick);
  run_op<OP_COPY>(out, buf[0], buf[1], buf[2], buf[3], sink, n2, n_cus, reps, quick);
  run_op<OP_TRIAD>(out, buf[0], buf[1], buf[2], buf[3], sink, n2, n_cus, reps, quick);
  run_op<OP_UPD1>(out, buf[0], buf[1], buf[2], buf[3], sink, n2, n_cus, reps, quick);
  run_op<OP_UPD2>(out, buf[0], buf[1], buf[2], buf[3], sink, n2, n_cus, reps, quick);
  printf("\n== best shape per op (median of %d launches) ==\n", reps);
  for (int op = 0; op < N_OPS; ++op) {
    const Result *best = nullptr, *lib = nullptr;
    for (const Result &r : out) {
      if (r.op != op) continue;
      if (!best || r.gbs > best->gbs) best = &r;
      if (r.U == (op >= OP_UPD1 ? 2 : 1) && !r.nt && r.vb == 256 && !strcmp(r.mode, "cyclic") && r.k == 8) lib = &r; // the library's shape (2048 blocks)
    }
    if (best)
      printf("%-5s best %7.1f GB/s  U=%d nt=%d VB=%d %s k=%d   | library-like shape (U=%d, cyclic, 2048 blocks): %7.1f GB/s\n", op_name[op], best->gbs, best->U,
             best->nt, best->vb, best->mode, best->k, op >= OP_UPD1 ? 2 : 1, lib ? lib->gbs : 0.0);
  }
  return 0;
}
