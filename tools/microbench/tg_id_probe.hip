// Which workgroups share a CU, and what HW_REG_HW_ID.TG_ID says about it (round 4: conv_tile_kernel<..., NW = 4> runs two 4-wave
// workgroups per CU and delays the one in the odd slot).  512 / 2048 workgroups of 256 threads with 76 KiB of LDS each (two fit a CU):
// every workgroup records (xcc, se, cu, tg_id, start, end).  Prints, per CU, the slots of its co-resident workgroups.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/tg_id_probe.hip -o tools/microbench/tg_id_probe && tools/microbench/tg_id_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>

struct Rec { unsigned hw, xcc; long long t0, t1; };

__global__ void __launch_bounds__(256, 2) probe(Rec* out, int spin) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();
  lds[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  const long long c0 = clock64();
  while (clock64() - c0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    Rec r;
    r.hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, all 32 bits
    r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    r.t0 = t0; r.t1 = wall_clock64();
    out[blockIdx.x] = r;
  }
}

int main() {
  for (int wgs : {512, 2048}) {
    Rec* d; hipMalloc(&d, sizeof(Rec) * wgs);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 76 * 1024, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<Rec> h(wgs);
    hipMemcpy(h.data(), d, sizeof(Rec) * wgs, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    for (int i = 0; i < wgs; ++i) {
      const unsigned hw = h[i].hw, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = h[i].xcc & 15;
      by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(i);
    }
    long long tmin = h[0].t0;
    for (auto& r : h) tmin = std::min(tmin, r.t0);
    std::printf("== %d workgroups: %zu distinct (xcc, se, sh, cu)\n", wgs, by_cu.size());
    int shown = 0, same_tg_overlap = 0, pairs = 0;
    for (auto& kv : by_cu) {
      auto& v = kv.second;
      // co-resident pairs: overlapping [t0, t1)
      for (size_t a = 0; a < v.size(); ++a)
        for (size_t b = a + 1; b < v.size(); ++b) {
          const Rec &A = h[v[a]], &B = h[v[b]];
          if (A.t0 < B.t1 && B.t0 < A.t1) { ++pairs; if (((A.hw >> 16) & 15) == ((B.hw >> 16) & 15)) ++same_tg_overlap; }
        }
      if (shown < 6) {
        std::printf("  cu %05x:", kv.first);
        for (int i : v) std::printf(" [wg %d tg %u wave %u simd %u t0 %lld t1 %lld]", i, (h[i].hw >> 16) & 15, h[i].hw & 15, (h[i].hw >> 4) & 3,
                                    h[i].t0 - tmin, h[i].t1 - tmin);
        std::printf("\n");
        ++shown;
      }
    }
    std::map<unsigned, int> tgc;
    for (auto& r : h) tgc[(r.hw >> 16) & 15]++;
    std::printf("  co-resident pairs %d, of which with EQUAL tg_id %d; tg_id histogram:", pairs, same_tg_overlap);
    for (auto& kv : tgc) std::printf(" %u:%d", kv.first, kv.second);
    std::printf("\n");
    hipFree(d);
  }
  return 0;
}
