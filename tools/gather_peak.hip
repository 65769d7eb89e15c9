// Microbenchmark: what the vector memory path of an MI355X CU delivers for per-lane gathers of small
// records from a cache-resident table — the access pattern of the closest-hit search (one 64-byte node,
// 48-byte triangle or 64-byte object record per lane and step, at unrelated addresses).
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_peak tools/gather_peak.hip && ./gather_peak > out.json
//
// Every lane draws record indices from its own hash sequence and reads the record with 16-byte loads.
// Variants: bytes per record read (16 / 48 / 64), how many lanes of a wave take part (the others are
// masked off by a branch, as lanes waiting for another kind of step are in k_extend), table size (L1-,
// L2-, Infinity-Cache-resident), and a quad-cooperative form in which four lanes read the four 16-byte
// pieces of one record (16 records per wave-instruction instead of 64).
// Output: one JSON object per line with the achieved record and byte rates.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
      std::exit(1);                                                                   \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// PIECES 16-byte loads per record; `active` lanes of each wave take part; QUAD: lanes 4k..4k+3 read the
// four pieces of the record that lane-group k draws (PIECES must be 4).  SHARE > 1: groups of SHARE
// neighbouring lanes draw the same record and each reads all of it (lanes of the search that stand at the
// same node: the L1 then serves one access for the group).
// `active` < 0: the -active lanes that take part are spread over the wave (lane numbers in bit-reversed order: 32
// lanes = every other lane, 16 lanes = one lane in every group of four) instead of being the lowest ones.
template <int PIECES, bool QUAD, int SHARE = 1>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ table, uint32_t nRecords, int iters,
                                                int active, float* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
  const uint32_t spread = __brev(lane) >> 26;  // 6-bit reversal
  if (active >= 0 ? (int)lane < active : (int)spread < -active) {
    uint32_t s = mix(QUAD ? (gid >> 2) + 1u : (gid / (uint32_t)SHARE) + 1u);
    constexpr int U = 4;  // independent records in flight per lane
    for (int it = 0; it < iters; it += U) {
      float4 v[U][QUAD ? 1 : PIECES];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s = s * 1664525u + 1013904223u;
        const uint32_t r = (uint32_t)(((uint64_t)mix(s) * nRecords) >> 32);
        const float4* p = table + (size_t)r * 4u;
        if (QUAD) {
          v[u][0] = p[lane & 3u];
        } else {
#pragma unroll
          for (int k = 0; k < PIECES; ++k) v[u][k] = p[k];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < (QUAD ? 1 : PIECES); ++k) acc += v[u][k].x + v[u][k].w;
    }
  }
  if (acc == 123.456f) out[gid] = acc;  // keeps the loads alive
}

struct Case {
  const char* name;
  int pieces;
  bool quad;
  int share;
};

int main(int argc, char** argv) {
  const bool spreadOnly = argc > 1 && std::string(argv[1]) == "spread";
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  const double clockGHz = prop.clockRate / 1e6;
  const size_t maxBytes = 256u << 20;
  float4* table = nullptr;
  float* out = nullptr;
  CHECK(hipMalloc(&table, maxBytes));
  CHECK(hipMemset(table, 0, maxBytes));
  const int wavesPerSimd = 8;
  const int blocks = cus * wavesPerSimd;  // 256 threads = 4 waves per block, one per SIMD
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const Case cases[] = {{"lane_64B", 4, false, 1},       {"lane_48B", 3, false, 1},       {"lane_16B", 1, false, 1},
                        {"quad_64B", 4, true, 1},        {"lane_64B_share2", 4, false, 2}, {"lane_64B_share4", 4, false, 4},
                        {"lane_64B_share8", 4, false, 8}};
  const size_t sizes[] = {16u << 10, 1u << 20, 2u << 20, 16u << 20, 48u << 20};
  // "spread": only the plain 64-byte records, lowest lanes against lanes spread over the wave
  const std::vector<int> actives = spreadOnly ? std::vector<int>{64, 32, -32, 16, -16, 8, -8} : std::vector<int>{64, 48, 32, 16};
  for (const Case& c : cases)
    for (size_t bytes : sizes)
      for (int active : actives) {
        if (spreadOnly && (c.pieces != 4 || c.quad || c.share != 1 || bytes > (2u << 20))) continue;
        const uint32_t nRec = (uint32_t)(bytes / 64);
        const int iters = 2048;
        auto launch = [&]() {
          if (c.quad) hipLaunchKernelGGL((k_gather<4, true>), dim3(blocks), dim3(256), 0, 0, table, nRec, iters, active, out);
          else if (c.share == 2) hipLaunchKernelGGL((k_gather<4, false, 2>), dim3(blocks), dim3(256), 0, 0, table, nRec, iters, active, out);
          else if (c.share == 4) hipLaunchKernelGGL((k_gather<4, false, 4>), dim3(blocks), dim3(256), 0, 0, table, nRec, iters, active, out);
          else if (c.share == 8) hipLaunchKernelGGL((k_gather<4, false, 8>), dim3(blocks), dim3(256), 0, 0, table, nRec, iters, active, out);
          else if (c.pieces == 4) hipLaunchKernelGGL((k_gather<4, false>), dim3(blocks), dim3(256), 0, 0, table, nRec, iters, active, out);
          else if (c.pieces == 3) hipLaunchKernelGGL((k_gather<3, false>), dim3(blocks), dim3(256), 0, 0, table, nRec, iters, active, out);
          else hipLaunchKernelGGL((k_gather<1, false>), dim3(blocks), dim3(256), 0, 0, table, nRec, iters, active, out);
        };
        launch();  // warm the caches
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double waves = (double)blocks * 4.0;
        // records read: per-lane forms one record per active lane and iteration; the quad form one per four lanes
        const int nActive = active < 0 ? -active : active;
        const double recs = waves * iters * (c.quad ? nActive / 4.0 : (double)nActive);
        const double bytesRead = recs * 16.0 * c.pieces;
        const double waveInstr = waves * iters * (c.quad ? 1.0 : (double)c.pieces);
        std::printf("{\"case\": \"%s\", \"table_bytes\": %zu, \"active_lanes\": %d, \"spread\": %d, \"ms\": %.4f, \"records_per_s\": %.4g, "
                    "\"gbytes_per_s\": %.1f, \"bytes_per_clk_per_cu\": %.2f, \"clk_per_wave_instr_per_cu\": %.2f, "
                    "\"cus\": %d, \"clock_ghz\": %.3f}\n",
                    c.name, bytes, nActive, active < 0 ? 1 : 0, ms, recs / (ms * 1e-3), bytesRead / (ms * 1e-3) / 1e9,
                    bytesRead / (ms * 1e-3) / (cus * clockGHz * 1e9), (ms * 1e-3) * clockGHz * 1e9 * cus / waveInstr, cus,
                    clockGHz);
        std::fflush(stdout);
      }
  return 0;
}
