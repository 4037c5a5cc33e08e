// test_shard.cpp -- the native batched-frames mode (mof_shard_*, include/mof.h) from a C++ host, no Python:
// one process, one engine + stream per device, contiguous ceil(B / G) shards, ONE in-place RCCL all-gather of the result
// slabs. Runs with the devices the box has (1 on the test pool; the code path is the same for 8) and checks every device's
// gathered result, bit for bit, against the single-engine call on the whole batch.
//   usage: test_shard <n_pairs> [n_devices]      prints "shard ok <devices> <pairs>" on success
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mof.h"
#include "mof/processors.hpp"

#define CHECK(x)                                                                     \
  do {                                                                               \
    if (!(x)) {                                                                      \
      std::fprintf(stderr, "FAILED %s:%d %s | %s\n", __FILE__, __LINE__, #x, mof_last_error()); \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

static uint32_t mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? std::atoi(argv[1]) : 37;
  const int have = mof_device_count();
  CHECK(have >= 1);
  const int G = argc > 2 ? std::atoi(argv[2]) : have;
  CHECK(G >= 1 && G <= have);
  // partition arithmetic: contiguous, ceil(B / G), covers [0, B) exactly once
  for (int g : {1, 2, 4, 8})
    for (int b : {0, 1, 7, 8, 37, 1000, 1024, 8192}) {
      int next = 0;
      for (int s = 0; s < g; ++s) {
        int first = -1, count = -1;
        CHECK(mof_shard_partition(b, g, s, &first, &count) == MOF_OK);
        CHECK(count >= 0 && count <= mof_shard_slab_pairs(b, g));
        CHECK(count == 0 || first == next);
        next += count;
      }
      CHECK(next == b);
    }
  CHECK(mof_shard_partition(8, 0, 0, nullptr, nullptr) == MOF_ERR_BAD_ARG);

  const int W = 200, H = 136, N = 64;
  mof_fft_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.frame_width = W; cfg.frame_height = H; cfg.patch_size = N;
  cfg.grid_x = 2; cfg.grid_y = 2; cfg.origin_x = 3; cfg.origin_y = 1; cfg.stride_x = 97; cfg.stride_y = 59;
  cfg.max_px_speed = 80.0; cfg.search_radius = 55;
  const size_t fb = (size_t)W * H, per_pair = (size_t)cfg.grid_x * cfg.grid_y * 2;
  // synthetic frames: pair k = a noise image and the same image rolled by (k % 5 - 2, k % 3 - 1)
  std::vector<uint8_t> cur(fb * B), prev(fb * B);
  for (int k = 0; k < B; ++k) {
    const int dx = k % 5 - 2, dy = k % 3 - 1;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        prev[fb * k + (size_t)y * W + x] = (uint8_t)(mix(0x5eedu + 977u * k + 65537u * (uint32_t)y + (uint32_t)x) >> 24);
        const int sy = (y - dy + H) % H, sx = (x - dx + W) % W;
        cur[fb * k + (size_t)y * W + x] = (uint8_t)(mix(0x5eedu + 977u * k + 65537u * (uint32_t)sy + (uint32_t)sx) >> 24);
      }
  }
  // reference: ONE engine on device 0, the whole batch
  std::vector<double> want(per_pair * B);
  {
    mof_fft_engine* e = nullptr;
    CHECK(mof_fft_create(&cfg, &e) == MOF_OK);
    CHECK(mof_fft_process_batch_host(e, cur.data(), fb, prev.data(), fb, W, B, want.data()) == MOF_OK);
    mof_fft_destroy(e);
  }
  // the sharded group: every device gets ITS shard of the frames and a full-size result buffer
  mof_shard_fft* g = nullptr;
  CHECK(mof_shard_fft_create(&cfg, nullptr, G, &g) == MOF_OK && mof_shard_fft_devices(g) == G);
  const int slab = mof_shard_slab_pairs(B, G);
  std::vector<uint8_t*> dc(G, nullptr), dp(G, nullptr);
  std::vector<double*> dout(G, nullptr);
  for (int s = 0; s < G; ++s) {
    int first = 0, count = 0;
    CHECK(mof_shard_partition(B, G, s, &first, &count) == MOF_OK);
    CHECK(hipSetDevice(s) == hipSuccess);
    CHECK(hipMalloc(&dout[s], sizeof(double) * per_pair * (size_t)slab * G) == hipSuccess);
    CHECK(hipMemset(dout[s], 0xff, sizeof(double) * per_pair * (size_t)slab * G) == hipSuccess);
    if (count > 0) {
      CHECK(hipMalloc(&dc[s], fb * count) == hipSuccess && hipMalloc(&dp[s], fb * count) == hipSuccess);
      CHECK(hipMemcpy(dc[s], cur.data() + fb * first, fb * count, hipMemcpyHostToDevice) == hipSuccess);
      CHECK(hipMemcpy(dp[s], prev.data() + fb * first, fb * count, hipMemcpyHostToDevice) == hipSuccess);
    }
  }
  for (int rep = 0; rep < 2; ++rep) {  // (the second batch re-uses the communicators)
    CHECK(mof_shard_fft_process_batch_device(g, (const uint8_t* const*)dc.data(), fb, (const uint8_t* const*)dp.data(), fb, W, B,
                                             dout.data(), /*gather=*/1) == MOF_OK);
    CHECK(mof_shard_fft_sync(g) == MOF_OK);
  }
  std::vector<double> got(per_pair * (size_t)slab * G);
  for (int s = 0; s < G; ++s) {
    CHECK(hipSetDevice(s) == hipSuccess);
    CHECK(hipMemcpy(got.data(), dout[s], sizeof(double) * got.size(), hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(std::memcmp(got.data(), want.data(), sizeof(double) * per_pair * B) == 0);  // every device holds ALL results, same bits
  }
  for (int s = 0; s < G; ++s) {
    (void)hipSetDevice(s);
    (void)hipFree(dc[s]); (void)hipFree(dp[s]); (void)hipFree(dout[s]);
  }
  mof_shard_fft_destroy(g);
  // the C++ mirror (include/mof/processors.hpp) on the same data: shard 0 of a group, no gather (each device keeps its own slab)
  {
    mof::ShardedFftMethod sm(cfg, G);
    CHECK(sm.devices() == G && sm.resultDoubles(B) == per_pair * (size_t)slab * G);
    int first = 0, count = 0;
    sm.partition(B, 0, &first, &count);
    CHECK(first == 0 && count == (B < slab ? B : slab));
  }
  std::printf("shard ok %d %d\n", G, B);
  return 0;
}
