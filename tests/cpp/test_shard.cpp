// test_shard.cpp -- the native batched-frames mode (mof_shard_*, include/mof.h) from a C++ host, no Python:
// one process, one engine + stream per shard, contiguous ceil(B / G) shards, ONE in-place RCCL all-gather of the result slabs.
//   test_shard <n_pairs> [n_devices]   the real group on the devices the box has (1 on the test pool; the code path is the same for
//                                      8): every device's gathered result, bit for bit, against the single-engine call on the
//                                      whole batch -- FftMethod and FastSpacedBMMethod (dx, dy and mode in one slab).
//                                      prints "shard ok <devices> <pairs>"
//   test_shard rehearse <n_pairs> <G>  MOF_SHARD_SHARE_DEVICE=1 must be set: G shards all on device 0, gather = 0 -- the G > 1 slab
//                                      arithmetic (slab i at i * slab, ragged and empty last shards, B < G) on a one-GPU box.
//                                      The gather itself cannot run this way (RCCL wants one rank per device): it is refused, and
//                                      stays a 1-rank run until a multi-GPU node exists.   prints "rehearse ok <G> <pairs>"
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

namespace {

constexpr int W = 200, H = 136;

// synthetic frames: pair k = a noise image and the same image rolled by (k % 5 - 2, k % 3 - 1)
void make_frames(int B, std::vector<uint8_t>& cur, std::vector<uint8_t>& prev) {
  const size_t fb = (size_t)W * H;
  cur.resize(fb * B);
  prev.resize(fb * B);
  for (int k = 0; k < B; ++k) {
    const int dx = k % 5 - 2, dy = k % 3 - 1;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        prev[fb * k + (size_t)y * W + x] = (uint8_t)(mix(0x5eedu + 977u * k + 65537u * (uint32_t)y + (uint32_t)x) >> 24);
        const int sy = (y - dy + H) % H, sx = (x - dx + W) % W;
        cur[fb * k + (size_t)y * W + x] = (uint8_t)(mix(0x5eedu + 977u * k + 65537u * (uint32_t)sy + (uint32_t)sx) >> 24);
      }
  }
}

mof_fft_config fft_cfg() {
  mof_fft_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.frame_width = W; cfg.frame_height = H; cfg.patch_size = 64;
  cfg.grid_x = 2; cfg.grid_y = 2; cfg.origin_x = 3; cfg.origin_y = 1; cfg.stride_x = 97; cfg.stride_y = 59;
  cfg.max_px_speed = 80.0; cfg.search_radius = 55;
  return cfg;
}

// every shard's frames on its device, plus a full-size result buffer filled with 0xff
struct Buffers {
  std::vector<uint8_t*> dc, dp;
  std::vector<unsigned char*> dout;
  std::vector<int> dev;
  ~Buffers() {
    for (size_t s = 0; s < dout.size(); ++s) {
      (void)hipSetDevice(dev[s]);
      (void)hipFree(dc[s]); (void)hipFree(dp[s]); (void)hipFree(dout[s]);
    }
  }
  int fill(const std::vector<int>& devices, int B, const std::vector<uint8_t>& cur, const std::vector<uint8_t>& prev, size_t out_bytes) {
    const int G = (int)devices.size();
    const size_t fb = (size_t)W * H;
    dev = devices;
    dc.assign(G, nullptr); dp.assign(G, nullptr); dout.assign(G, nullptr);
    for (int s = 0; s < G; ++s) {
      int first = 0, count = 0;
      CHECK(mof_shard_partition(B, G, s, &first, &count) == MOF_OK);
      CHECK(hipSetDevice(devices[s]) == hipSuccess);
      CHECK(hipMalloc(&dout[s], out_bytes ? out_bytes : 16) == hipSuccess);
      CHECK(hipMemset(dout[s], 0xff, out_bytes ? out_bytes : 16) == hipSuccess);
      if (count > 0) {
        CHECK(hipMalloc(&dc[s], fb * count) == hipSuccess && hipMalloc(&dp[s], fb * count) == hipSuccess);
        CHECK(hipMemcpy(dc[s], cur.data() + fb * first, fb * count, hipMemcpyHostToDevice) == hipSuccess);
        CHECK(hipMemcpy(dp[s], prev.data() + fb * first, fb * count, hipMemcpyHostToDevice) == hipSuccess);
      }
    }
    return 0;
  }
};

// FftMethod group against one engine on the whole batch. gather: every device must hold ALL results; otherwise device s holds its
// own slab at s * slab and the 0xff fill everywhere else (nothing written outside a rank's own pairs).
int run_fft(const std::vector<int>& devices, int B, bool gather) {
  const int G = (int)devices.size();
  const mof_fft_config cfg = fft_cfg();
  const size_t fb = (size_t)W * H, per_pair = (size_t)cfg.grid_x * cfg.grid_y * 2;
  std::vector<uint8_t> cur, prev;
  make_frames(B, cur, prev);
  std::vector<double> want(per_pair * B);
  {
    mof_fft_engine* e = nullptr;
    CHECK(mof_fft_create(&cfg, &e) == MOF_OK);
    CHECK(mof_fft_process_batch_host(e, cur.data(), fb, prev.data(), fb, W, B, want.data()) == MOF_OK);
    mof_fft_destroy(e);
  }
  mof_shard_fft* g = nullptr;
  CHECK(mof_shard_fft_create(&cfg, devices.data(), G, &g) == MOF_OK && mof_shard_fft_devices(g) == G);
  const int slab = mof_shard_slab_pairs(B, G);
  const size_t out_doubles = per_pair * (size_t)slab * G;
  Buffers b;
  if (b.fill(devices, B, cur, prev, sizeof(double) * out_doubles)) return 1;
  std::vector<double*> dout(G);
  for (int s = 0; s < G; ++s) dout[s] = reinterpret_cast<double*>(b.dout[s]);
  CHECK(hipSetDevice(devices[0]) == hipSuccess);
  int before = -1, after = -1;
  CHECK(hipGetDevice(&before) == hipSuccess);
  if (gather) {
    // the contract: the asynchronous call never builds communicators -- without init_gather it refuses
    CHECK(mof_shard_fft_gather_ready(g) == 0);
    CHECK(mof_shard_fft_process_batch_device(g, (const uint8_t* const*)b.dc.data(), fb, (const uint8_t* const*)b.dp.data(), fb, W, B,
                                             dout.data(), 1) == MOF_ERR_NOT_INIT);
    CHECK(mof_shard_fft_init_gather(g) == MOF_OK && mof_shard_fft_gather_ready(g) == 1);
    CHECK(mof_shard_fft_init_gather(g) == MOF_OK);  // idempotent
  }
  for (int rep = 0; rep < 2; ++rep) {  // (the second batch re-uses the communicators)
    CHECK(mof_shard_fft_process_batch_device(g, (const uint8_t* const*)b.dc.data(), fb, (const uint8_t* const*)b.dp.data(), fb, W, B,
                                             dout.data(), gather ? 1 : 0) == MOF_OK);
    CHECK(mof_shard_fft_sync(g) == MOF_OK);
  }
  CHECK(hipGetDevice(&after) == hipSuccess && after == before);  // the caller's current device is the caller's
  std::vector<double> got(out_doubles ? out_doubles : 1);
  std::vector<unsigned char> ff(sizeof(double) * per_pair, 0xff);
  for (int s = 0; s < G; ++s) {
    CHECK(hipSetDevice(devices[s]) == hipSuccess);
    CHECK(hipMemcpy(got.data(), dout[s], sizeof(double) * out_doubles, hipMemcpyDeviceToHost) == hipSuccess);
    if (gather) {
      CHECK(std::memcmp(got.data(), want.data(), sizeof(double) * per_pair * B) == 0);  // every device holds ALL results, same bits
    } else {
      int first = 0, count = 0;
      CHECK(mof_shard_partition(B, G, s, &first, &count) == MOF_OK);
      CHECK(count == 0 || first == s * slab);  // slab s starts at pair index s * slab
      for (int k = 0; k < slab * G; ++k) {
        const bool mine = k >= first && k < first + count;
        const void* ref = mine ? (const void*)(want.data() + per_pair * k) : (const void*)ff.data();
        CHECK(std::memcmp(got.data() + per_pair * k, ref, sizeof(double) * per_pair) == 0);
      }
    }
  }
  // a missing frame pointer of a LATER shard is caught before anything is launched
  int last_first = 0, last_count = 0;
  CHECK(mof_shard_partition(B, G, G - 1, &last_first, &last_count) == MOF_OK);
  if (G > 1 && last_count > 0) {  // (an EMPTY last shard carries no frames: a null pointer is legal there)
    std::vector<uint8_t*> dc2 = b.dc;
    dc2[G - 1] = nullptr;
    CHECK(mof_shard_fft_process_batch_device(g, (const uint8_t* const*)dc2.data(), fb, (const uint8_t* const*)b.dp.data(), fb, W, B,
                                             dout.data(), 0) == MOF_ERR_BAD_ARG);
  }
  mof_shard_fft_destroy(g);
  return 0;
}

// FastSpacedBMMethod group (dx | dy | mode planes in one slab) against one engine on the whole batch
int run_bm(const std::vector<int>& devices, int B, bool gather) {
  const int G = (int)devices.size();
  mof_bm_config cfg;
  CHECK(mof_bm_config_fast_spaced(&cfg, W, H, 16, 8, 8) == MOF_OK);
  const size_t fb = (size_t)W * H, blocks = (size_t)cfg.grid_x * cfg.grid_y;
  std::vector<uint8_t> cur, prev;
  make_frames(B, cur, prev);
  std::vector<int8_t> wdx(blocks * B), wdy(blocks * B), wmode(8 * (size_t)B);
  {
    mof_bm_engine* e = nullptr;
    CHECK(mof_bm_create(&cfg, &e) == MOF_OK);
    CHECK(mof_bm_process_batch_host(e, cur.data(), fb, prev.data(), fb, W, B, wdx.data(), wdy.data(), wmode.data()) == MOF_OK);
    mof_bm_destroy(e);
  }
  mof_shard_bm* g = nullptr;
  CHECK(mof_shard_bm_create(&cfg, devices.data(), G, &g) == MOF_OK && mof_shard_bm_devices(g) == G);
  const size_t slab = mof_shard_bm_slab_bytes(g, B), sp = (size_t)mof_shard_slab_pairs(B, G);
  CHECK(slab % 16 == 0 && slab >= sp * (2 * blocks + 8));
  Buffers b;
  if (b.fill(devices, B, cur, prev, slab * G)) return 1;
  std::vector<int8_t*> dout(G);
  for (int s = 0; s < G; ++s) dout[s] = reinterpret_cast<int8_t*>(b.dout[s]);
  if (gather) {
    CHECK(mof_shard_bm_process_batch_device(g, (const uint8_t* const*)b.dc.data(), fb, (const uint8_t* const*)b.dp.data(), fb, W, B,
                                            dout.data(), 1) == MOF_ERR_NOT_INIT);
    CHECK(mof_shard_bm_init_gather(g) == MOF_OK && mof_shard_bm_gather_ready(g) == 1);
  }
  CHECK(mof_shard_bm_process_batch_device(g, (const uint8_t* const*)b.dc.data(), fb, (const uint8_t* const*)b.dp.data(), fb, W, B,
                                          dout.data(), gather ? 1 : 0) == MOF_OK);
  CHECK(mof_shard_bm_sync(g) == MOF_OK);
  std::vector<int8_t> got(slab * G);
  for (int s = 0; s < G; ++s) {
    CHECK(hipSetDevice(devices[s]) == hipSuccess);
    CHECK(hipMemcpy(got.data(), dout[s], slab * G, hipMemcpyDeviceToHost) == hipSuccess);
    int first = 0, count = 0;
    CHECK(mof_shard_partition(B, G, s, &first, &count) == MOF_OK);
    for (int k = 0; k < B; ++k) {
      size_t ox = 0, oy = 0, om = 0;
      CHECK(mof_shard_bm_locate(g, B, k, &ox, &oy, &om) == MOF_OK);
      CHECK(ox / slab == (size_t)k / sp && om + 8 <= slab * G);
      const bool here = gather || (k >= first && k < first + count);
      if (here) {
        CHECK(std::memcmp(got.data() + ox, wdx.data() + blocks * k, blocks) == 0);
        CHECK(std::memcmp(got.data() + oy, wdy.data() + blocks * k, blocks) == 0);
        CHECK(std::memcmp(got.data() + om, wmode.data() + 8 * (size_t)k, 8) == 0);  // the mode rides in the same slab
      } else {
        for (size_t i = 0; i < blocks; ++i) CHECK((uint8_t)got[ox + i] == 0xff && (uint8_t)got[oy + i] == 0xff);
      }
    }
  }
  CHECK(mof_shard_bm_locate(g, B, B, nullptr, nullptr, nullptr) == MOF_ERR_BAD_ARG);
  mof_shard_bm_destroy(g);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  const bool rehearse = argc > 1 && std::string(argv[1]) == "rehearse";
  const int have = mof_device_count();
  CHECK(have >= 1);
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

  if (rehearse) {
    CHECK(argc > 3);
    const int B = std::atoi(argv[2]), G = std::atoi(argv[3]);
    CHECK(B >= 1 && G >= 2 && G <= 16);
    const char* knob = getenv("MOF_SHARD_SHARE_DEVICE");
    CHECK(knob && std::atoi(knob) != 0);
    std::vector<int> devices(G, 0);  // every shard on device 0
    if (run_fft(devices, B, /*gather=*/false)) return 1;
    if (run_bm(devices, B, /*gather=*/false)) return 1;
    // such a group cannot gather, and says so
    const mof_fft_config cfg = fft_cfg();
    mof_shard_fft* g = nullptr;
    CHECK(mof_shard_fft_create(&cfg, devices.data(), G, &g) == MOF_OK);
    CHECK(mof_shard_fft_init_gather(g) == MOF_ERR_UNSUPPORTED && mof_shard_fft_gather_ready(g) == 0);
    mof_shard_fft_destroy(g);
    std::printf("rehearse ok %d %d\n", G, B);
    return 0;
  }

  const int B = argc > 1 ? std::atoi(argv[1]) : 37;
  const int G = argc > 2 ? std::atoi(argv[2]) : have;
  CHECK(G >= 1 && G <= have);
  std::vector<int> devices(G);
  for (int i = 0; i < G; ++i) devices[i] = i;
  // without the knob a device may carry one shard only
  if (!getenv("MOF_SHARD_SHARE_DEVICE")) {
    const mof_fft_config cfg = fft_cfg();
    const int twice[2] = {0, 0};
    mof_shard_fft* g = nullptr;
    CHECK(mof_shard_fft_create(&cfg, twice, 2, &g) == MOF_ERR_BAD_ARG && g == nullptr);
  }
  if (run_fft(devices, B, /*gather=*/true)) return 1;
  if (run_fft(devices, B, /*gather=*/false)) return 1;
  if (run_bm(devices, B, /*gather=*/true)) return 1;
  // the C++ mirrors (include/mof/processors.hpp) on the same geometry
  {
    const mof_fft_config cfg = fft_cfg();
    mof::ShardedFftMethod sm(cfg, G);
    const int slab = mof_shard_slab_pairs(B, G);
    CHECK(sm.devices() == G && sm.resultDoubles(B) == (size_t)cfg.grid_x * cfg.grid_y * 2 * (size_t)slab * G);
    int first = 0, count = 0;
    sm.partition(B, 0, &first, &count);
    CHECK(first == 0 && count == (B < slab ? B : slab));
    CHECK(!sm.gatherReady());
    sm.initGather();
    CHECK(sm.gatherReady());
    mof_bm_config bc;
    CHECK(mof_bm_config_fast_spaced(&bc, W, H, 16, 8, 8) == MOF_OK);
    mof::ShardedBlockMatcher sb(bc, G);
    CHECK(sb.devices() == G && sb.resultBytes(B) % 16 == 0);
    const mof::ShardedBlockMatcher::Where w = sb.locate(B, B - 1);
    CHECK(w.mode + 8 <= sb.resultBytes(B) && w.dx < w.dy && w.dy < w.mode);
  }
  std::printf("shard ok %d %d\n", G, B);
  return 0;
}
