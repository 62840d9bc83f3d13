// Cost of the "last workgroup finishes the job" pattern on gfx950 (tuning aid): every workgroup writes a slab, then
//   A: nothing more                                   B: __threadfence + atomic ticket, nobody reads
//   C: B + the last workgroup reads ONE float per workgroup written by the others (a norm-finalize sized tail)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/last_arrival.hip -o tools/microbench/last_arrival
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(512) k(float* slab, float* sums, int* counter, float* out, int slab_floats) {
  const int t = threadIdx.x, b = blockIdx.x;
  float acc = 0.f;
  for (int i = t; i < slab_floats; i += 512) {
    const float v = (float)(i + b);
    slab[(size_t)b * slab_floats + i] = v;
    acc += v;
  }
  // per-workgroup partial (like the per-tile moments of a convolution)
  __shared__ float red[512];
  red[t] = acc;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int i = 0; i < 512; ++i) s += red[i];
    sums[b] = s;
  }
  if (MODE == 0) return;
  __shared__ int last;
  __threadfence();
  __syncthreads();
  if (t == 0) last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  if (t == 0) *counter = 0;
  if (MODE == 1) return;
  __threadfence();
  float s = 0.f;
  for (int i = t; i < (int)gridDim.x; i += 512) s += sums[i];
  red[t] = s;
  __syncthreads();
  if (t == 0) {
    float tot = 0.f;
    for (int i = 0; i < 512; ++i) tot += red[i];
    *out = tot;
  }
}

template <int MODE>
float run(int blocks, int slab_floats) {
  float *slab, *sums, *out; int* counter;
  hipMalloc(&slab, (size_t)blocks * slab_floats * 4); hipMalloc(&sums, blocks * 4); hipMalloc(&out, 4); hipMalloc(&counter, 4);
  hipMemset(counter, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) k<MODE><<<blocks, 512>>>(slab, sums, counter, out, slab_floats);
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) k<MODE><<<blocks, 512>>>(slab, sums, counter, out, slab_floats);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(slab); hipFree(sums); hipFree(out); hipFree(counter);
  return ms * 1000.f / 200.f;
}

int main() {
  for (int blocks : {16, 64, 256}) for (int kb : {4, 64}) {
    const int fl = kb * 256;
    printf("blocks %3d slab %3d KiB:  plain %.2f us   fence+ticket %.2f us   fence+ticket+tail %.2f us\n", blocks, kb,
           run<0>(blocks, fl), run<1>(blocks, fl), run<2>(blocks, fl));
  }
  return 0;
}
