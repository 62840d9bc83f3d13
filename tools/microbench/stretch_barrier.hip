// Go / no-go measurement for a persistent "small-map stretch" kernel (round-4 review, task 1): what does ONE layer boundary cost when the
// layers of a 16x16 512 -> 512 3x3 stretch stay inside one launch, against one launch per layer?
//
// Every workgroup (256 of 512 threads, one per CU) plays (output block bo, pixel tile) of such a layer with the MEMORY behaviour of
// conv_small_kernel and none of its arithmetic:
//   produce : wave 0 stores its 32 px x 16 ch fp32 outputs (2 KiB) + 128 B of per-tile moments
//   boundary: kernel boundary | grid barrier (variants below)
//   consume : every thread reads its share of the moments of ALL 512 channels x 8 tiles (32 KiB per workgroup) and of the operand
//             window (60 px x 512 ch fp32 = 120 KiB per workgroup), and CHECKS every value (layer tag), i.e. stale data is counted.
// Modes 0-5 carry no weight stream (it is independent of the boundary); modes 6-8 add it - 288 KiB per workgroup and layer straight from L2 / HBM into
// registers, two 9-tap units per wave like conv_small_kernel - to price the one thing only a persistent kernel can do: request the NEXT layer's
// first unit before the boundary.
// Variants (MODE):
//   0 one launch per layer (the shipped form: boundary = kernel boundary)
//   1 persistent, plain stores, lane-0 release fence -> flat counter -> acquire fence
//   2 persistent, plain stores, release fence, XCD-hierarchical counters, acquire fence
//   3 persistent, sc1 (write-through) stores, no release fence, flat counter, acquire fence
//   4 persistent, sc1 stores, hierarchical counters, acquire fence
//   5 persistent, sc1 stores AND sc1 loads, hierarchical counters, no fence at all
//   6 = 0 + weight stream        7 = 5 + weight stream, requested behind the barrier        8 = 5 + weight stream, next layer's first unit requested BEFORE the barrier
//   9 = 6 + L2 WARM-UP of the next layer's weights (round-5 review, task 3a): each workgroup touches one dword per 128-B line of ITS eighth of the next
//       layer's slice for its output block (36 KiB = 288 lines; the 8 pixel tiles of an output block share an XCD, so the whole 288-KiB slice is in that
//       XCD's L2 when the next launch asks for it) right behind its own first requests; the values are never used (one compare at the end of the kernel)
//  10 = 9 with the warm-up loads issued in FRONT of the layer's own first requests (in-order vmcnt: do they delay the layer's own data?)
// Prints us per layer and the number of stale values seen.  Every spin is bounded (a stuck barrier sets a flag and the kernel leaves).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/stretch_barrier.hip -o tools/microbench/stretch_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512;
constexpr int kNB = 32, kTiles = 8, kWGs = kNB * kTiles;       // 16x16 map, 32 px per tile, 512 output channels
constexpr int kPx = 256, kCB = 32;                             // C16 activations [cb][px][16]
constexpr int kWinPx = 60;                                     // (4 + 2) x (8 + 2) window of a 4 x 8 tile

struct Params {
  float* act[2];        // ping-pong C16 tensors [32][256][16]
  float* stats[2];      // [8 tiles][512][2]
  unsigned* ctr;        // [0] flat counter, [32*(1+x)] per-XCD counters, [32*9] top counter, [32*10] generation, [32*11] timeout flag, [32*(12+x)] census
  unsigned long long* errors;
  const float* weights;  // [layer][bo 32][unit 16][tap 9][512 floats]: 9.4 MB per layer
  float* sink;
  int layers;
  int first_layer;      // MODE 0: the layer this launch plays
};

__device__ __forceinline__ void st16(float* p, f32x4 v, bool sc1) {
  if (sc1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *reinterpret_cast<f32x4*>(p) = v;
}
__device__ __forceinline__ unsigned ld_rlx(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__device__ __forceinline__ bool grid_barrier(const Params& p, int gen, int xcc, int nx, int nxcd) {
  // gen = 1, 2, ...: the number of barriers passed after this one
  if (MODE == 8) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");   // the 18 prefetched weight loads are YOUNGER than the stores (gfx9: vmcnt retires in order)
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its (write-through) stores
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    if (MODE == 1 || MODE == 2) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bool good = true;
    if (MODE == 1 || MODE == 3) {
      __hip_atomic_fetch_add(p.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)gen * gridDim.x;
      unsigned spins = 0;
      while (ld_rlx(p.ctr) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { good = false; break; }
      }
    } else {
      const unsigned old = __hip_atomic_fetch_add(p.ctr + 32 * (1 + xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == (unsigned)gen * (unsigned)nx) {           // last arrival of this XCD
        const unsigned t = __hip_atomic_fetch_add(p.ctr + 32 * 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == (unsigned)gen * (unsigned)nxcd) __hip_atomic_store(p.ctr + 32 * 10, (unsigned)gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      unsigned spins = 0;
      while (ld_rlx(p.ctr + 32 * 10) < (unsigned)gen) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { good = false; break; }
      }
    }
    if (MODE < 5) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!good) __hip_atomic_store(p.ctr + 32 * 11, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

// consume layer L-1's tensor (tag = L), produce layer L's (tag = L + 1)
struct WUnit { f32x4 v[18]; };
__device__ __forceinline__ void load_unit(const Params& p, int L, int bo, int unit, WUnit& w) {
  const float* base = p.weights + ((((size_t)L * 32 + bo) * 16 + unit) * 9) * 512 + (threadIdx.x & 63) * 4;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    w.v[2 * i] = *reinterpret_cast<const f32x4*>(base + i * 512);
    w.v[2 * i + 1] = *reinterpret_cast<const f32x4*>(base + i * 512 + 256);
  }
}
__device__ __forceinline__ float use_unit(const WUnit& w) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 18; ++i) s += (w.v[i][0] + w.v[i][1]) + (w.v[i][2] + w.v[i][3]);
  return s;
}

template <int MODE>
__device__ __forceinline__ void layer_body(const Params& p, int L, int b, WUnit& wpre) {
  constexpr bool kSc1St = MODE >= 3 && MODE != 6 && MODE < 9, kSc1Ld = MODE == 5 || MODE == 7 || MODE == 8;
  constexpr bool kW = MODE >= 6;
  constexpr bool kWarm = MODE >= 9;
  float warm = 0.f;
  // the next layer's slice for this output block: [L + 1][bo][16 units][9 taps][512 floats] = 288 KiB; this workgroup's eighth = 288 lines of 128 B
  const float* warm_ptr = p.weights + ((((size_t)(L + 1) * 32 + (b & 31)) * 16) * 9) * 512 + ((size_t)(b >> 5) * 288 + (threadIdx.x < 288 ? threadIdx.x : 0)) * 32;
  const bool do_warm = kWarm && L + 1 < p.layers && threadIdx.x < 288;
  if (MODE == 10 && do_warm) warm = *warm_ptr;
  WUnit w1;
  if (kW && MODE != 8) load_unit(p, L, b & 31, threadIdx.x >> 6, w1);        // first unit: requested with the window (top of the layer)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bo = b & 31, tile = b >> 5;
  const float* in = p.act[L & 1];
  const float* st = p.stats[L & 1];
  float* out = p.act[(L + 1) & 1];
  float* so = p.stats[(L + 1) & 1];
  const float tag = (float)L;
  unsigned long long bad = 0;
  // moments: 8 tiles x 512 ch x 2 floats = 8192 floats = 2048 f32x4 -> 4 per thread
  f32x4 m[4];
  // window: 60 px x 32 cb x 4 quads = 7680 f32x4 -> 15 per thread; pixel rows around this tile (wrapping: only the traffic matters)
  f32x4 w[15];
  if (kSc1Ld) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(m[i]) : "v"(st + (size_t)(t + i * kThreads) * 4) : "memory");
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      const int item = t + i * kThreads;                       // [cb 32][px 60][quad 4]
      const int q = item & 3, px = (item >> 2) % kWinPx, cb = (item >> 2) / kWinPx;
      const int gpx = (tile * 32 + px + 226) & 255;
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[i]) : "v"(in + ((size_t)cb * kPx + gpx) * 16 + q * 4) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = *reinterpret_cast<const f32x4*>(st + (size_t)(t + i * kThreads) * 4);
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      const int item = t + i * kThreads;
      const int q = item & 3, px = (item >> 2) % kWinPx, cb = (item >> 2) / kWinPx;
      const int gpx = (tile * 32 + px + 226) & 255;
      w[i] = *reinterpret_cast<const f32x4*>(in + ((size_t)cb * kPx + gpx) * 16 + q * 4);
    }
  }
  if (MODE == 9 && do_warm) warm = *warm_ptr;               // behind the layer's own first requests (moments, window, first weight unit)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) bad += m[i][j] != tag;
#pragma unroll
  for (int i = 0; i < 15; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) bad += w[i][j] != tag;
  if (bad) atomicAdd(p.errors, bad);
  if (kW) {
    float acc = use_unit(MODE == 8 ? wpre : w1);
    WUnit w2;
    load_unit(p, L, b & 31, 8 + (threadIdx.x >> 6), w2);     // second unit: requested when the first has been multiplied
    acc += use_unit(w2);
    if (acc == 12345.678f) p.sink[threadIdx.x] = acc;
  }
  // produce: wave 0 writes 32 px x 16 ch (lane = pixel column p, group g: 4 channels) x 2 pixel groups, and the tile's moments
  if (wave == 0) {
    const f32x4 v = {tag + 1.f, tag + 1.f, tag + 1.f, tag + 1.f};
    const int pcol = lane & 15, g = lane >> 4;
#pragma unroll
    for (int pg = 0; pg < 2; ++pg) st16(out + ((size_t)bo * kPx + tile * 32 + pg * 16 + pcol) * 16 + g * 4, v, kSc1St);
    if (lane < 8) st16(so + ((size_t)tile * 512 + bo * 16) * 2 + lane * 4, v, kSc1St);
  }
  if (MODE == 8 && L + 1 < p.layers) load_unit(p, L + 1, b & 31, threadIdx.x >> 6, wpre);   // next layer's first unit, in front of the barrier
  if (kWarm && warm == 12345.678f) p.sink[threadIdx.x] = warm;                              // (the only use of the warm-up values: nothing waits for them before this)
}

template <int MODE>
__global__ void __launch_bounds__(kThreads) stretch_kernel(Params p) {
  const int b = blockIdx.x;
  WUnit wpre;
  if (MODE == 0 || MODE == 6 || MODE >= 9) { layer_body<MODE>(p, p.first_layer, b, wpre); return; }
  // census: workgroups per XCD (established behind one flat, fenced barrier)
  const int xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) & 15;
  __shared__ int s_nx, s_nxcd;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(p.ctr + 32 * (12 + xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(p.ctr + 32 * 24, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (ld_rlx(p.ctr + 32 * 24) < gridDim.x && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
    int n = 0;
    for (int x = 0; x < 8; ++x) n += ld_rlx(p.ctr + 32 * (12 + x)) != 0;
    s_nx = (int)ld_rlx(p.ctr + 32 * (12 + xcc));
    s_nxcd = n;
  }
  __syncthreads();
  const int nx = s_nx, nxcd = s_nxcd;
  if (MODE == 8) load_unit(p, 0, b & 31, threadIdx.x >> 6, wpre);
  for (int L = 0; L < p.layers; ++L) {
    layer_body<MODE>(p, L, b, wpre);
    if (!grid_barrier<MODE>(p, L + 1, xcc, nx, nxcd)) return;
  }
}

template <int MODE>
void run(int layers, int reps) {
  Params p{};
  const size_t act_f = (size_t)kCB * kPx * 16, st_f = (size_t)kTiles * 512 * 2;
  for (int i = 0; i < 2; ++i) { hipMalloc(&p.act[i], act_f * 4); hipMalloc(&p.stats[i], st_f * 4); }
  hipMalloc(&p.ctr, 32 * 32 * 4);
  hipMalloc(&p.errors, 8);
  static float* weights = nullptr;                   // zeros: only the traffic matters
  static float* sink = nullptr;
  if (!weights) { hipMalloc(&weights, (size_t)layers * 32 * 16 * 9 * 512 * 4); hipMemset(weights, 0, (size_t)layers * 32 * 16 * 9 * 512 * 4); hipMalloc(&sink, 4096); }
  p.weights = weights; p.sink = sink;
  p.layers = layers;
  std::vector<float> zero(act_f, 0.f);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  unsigned long long errs_total = 0;
  unsigned timeout = 0;
  for (int r = 0; r < reps + 1; ++r) {
    // layer 0 consumes tag 0
    hipMemset(p.act[0], 0, act_f * 4); hipMemset(p.stats[0], 0, st_f * 4);
    hipMemset(p.act[1], 0xff, act_f * 4); hipMemset(p.stats[1], 0xff, st_f * 4);
    hipMemset(p.ctr, 0, 32 * 32 * 4); hipMemset(p.errors, 0, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    if (MODE == 0 || MODE == 6 || MODE >= 9) {
      for (int L = 0; L < layers; ++L) { p.first_layer = L; stretch_kernel<MODE><<<kWGs, kThreads>>>(p); }
    } else {
      stretch_kernel<MODE><<<kWGs, kThreads>>>(p);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
    unsigned long long e; unsigned tmo;
    hipMemcpy(&e, p.errors, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&tmo, p.ctr + 32 * 11, 4, hipMemcpyDeviceToHost);
    errs_total += e; timeout |= tmo;
  }
  static const char* names[] = {"0 one launch per layer", "1 persistent, release fence + flat counter + acquire", "2 persistent, release fence + XCD counters + acquire",
                                "3 persistent, sc1 stores + flat counter + acquire", "4 persistent, sc1 stores + XCD counters + acquire",
                                "5 persistent, sc1 stores + sc1 loads + XCD counters, no fence", "6 = 0 + 288 KiB weight stream per workgroup",
                                "7 = 5 + weight stream requested behind the barrier", "8 = 5 + next layer's first unit requested before the barrier",
                                "9 = 6 + next layer's weights warmed into L2 (behind own requests)", "10 = 6 + next layer's weights warmed into L2 (in front)"};
  std::printf("%-62s  %7.2f us / layer  (%d layers, best of %d)  stale values %llu%s\n", names[MODE], best * 1000.f / layers, layers, reps, errs_total,
              timeout ? "  BARRIER TIMED OUT" : "");
  for (int i = 0; i < 2; ++i) { hipFree(p.act[i]); hipFree(p.stats[i]); }
  hipFree(p.ctr); hipFree(p.errors);
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? std::atoi(argv[1]) : 44, reps = 5;
  run<0>(layers, reps);
  run<1>(layers, reps);
  run<2>(layers, reps);
  run<3>(layers, reps);
  run<4>(layers, reps);
  run<5>(layers, reps);
  run<6>(layers, reps);
  run<7>(layers, reps);
  run<8>(layers, reps);
  run<9>(layers, reps);
  run<10>(layers, reps);
  run<6>(layers, reps);
  return 0;
}
