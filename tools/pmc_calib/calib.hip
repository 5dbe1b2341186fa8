// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths the BA kernels use (8 B per lane, one
// workgroup streaming its own contiguous slice), as MI355X_MICROARCH.md asks before trusting an absolute.
// Each kernel moves a known byte count: read_f64 reads 1 GiB, write_f64 writes 1 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read_f64(const double* p, size_t n_per_block, double* out) {
  const double* q = p + (size_t)blockIdx.x * n_per_block;
  double s = 0;
  for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) s += q[i];
  if (s == 123.456) out[0] = s;
}
__global__ void read_f64x2(const double2* p, size_t n_per_block, double* out) {
  const double2* q = p + (size_t)blockIdx.x * n_per_block;
  double s = 0;
  for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) { double2 v = q[i]; s += v.x + v.y; }
  if (s == 123.456) out[0] = s;
}
__global__ void write_f64(double* p, size_t n_per_block) {
  double* q = p + (size_t)blockIdx.x * n_per_block;
  for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) q[i] = (double)i;
}
int main() {
  const size_t bytes = 1ull << 30, n = bytes / 8;
  double *a, *o;
  hipMalloc(&a, bytes); hipMalloc(&o, 64);
  hipMemset(a, 0, bytes);
  const int blocks = 512;
  hipLaunchKernelGGL(write_f64, dim3(blocks), dim3(512), 0, 0, a, n / blocks);
  hipLaunchKernelGGL(read_f64, dim3(blocks), dim3(512), 0, 0, a, n / blocks, o);
  hipLaunchKernelGGL(read_f64x2, dim3(blocks), dim3(512), 0, 0, (const double2*)a, n / 2 / blocks, o);
  hipDeviceSynchronize();
  std::printf("moved %zu bytes per kernel\n", bytes);
  return 0;
}
