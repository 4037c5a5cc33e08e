// mof_shard.hip -- the batched-frames mode across the GPUs of one node, natively behind the C ABI (mof_shard_* in include/mof.h).
//
// BASELINE north_star / SURVEY section 8(e): frame pairs are independent, so a batch of B pairs is cut into contiguous shards of
// ceil(B / G) pairs, shard g lives on (and is processed by) device g -- no data-path collective -- and ONE all-gather of the
// per-rank result slabs (RCCL over xGMI) hands every device the whole result. The host side stays C++ (the reference is a C++
// ROS node) and needs no Python: ONE process, one engine + one HIP stream per device, ncclCommInitAll for the communicators
// (explicit: mof_shard_*_init_gather), ncclGroupStart / ncclAllGather (in place) / ncclGroupEnd per batch.
// RCCL is bound at run time (dlopen of librccl.so.1 by init_gather): a single-GPU host -- the ROS node -- never maps the 570 MB
// library, and a process that already holds a copy (PyTorch bundles one under the same SONAME) shares it.
// One group core (devices, streams, communicators, the gather) serves both engine kinds:
//   mof_shard_fft_*  FftMethod: slab = ceil(B/G) * grid * 2 doubles                   (per device mof_fft_process_batch_device)
//   mof_shard_bm_*   BlockMethod / FastSpacedBMMethod: slab = dx | dy | mode planes   (per device mof_bm_process_batch_device;
//                    "BM mode vectors ride in the same slab", SURVEY section 8(e), the output pair of
//                    /root/reference/src/FastSpacedBMMethod_OCL.cpp:172-175)
// The reference has no counterpart (one synchronous call per frame on one device).

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "mof.h"

namespace mof {
int capi_fail(int code, const char* fmt, ...);  // mof_capi.hip
}

namespace {

// the handful of RCCL entry points the gather needs (signatures: /opt/rocm/include/rccl/rccl.h)
typedef void* rcclComm_t;
struct Rccl {
  int (*CommInitAll)(rcclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*CommCount)(const rcclComm_t, int*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int /*ncclDataType_t*/, rcclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
  char why[256] = "";
};
constexpr int kNcclInt8 = 0;  // ncclInt8 / ncclChar: the slabs travel as bytes (doubles and int8 results alike)

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
      if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (!r.handle) {
      snprintf(r.why, sizeof(r.why), "%s", dlerror());
      return;
    }
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(r.handle, name);
      if (!p && !r.why[0]) snprintf(r.why, sizeof(r.why), "librccl lacks %s", name);
      return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return (r.handle && !r.why[0]) ? &r : nullptr;
}

// the calling thread's current device is the caller's business: every entry point that switches devices puts it back
struct DeviceRestore {
  int dev = -1;
  DeviceRestore() {
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  }
  ~DeviceRestore() {
    if (dev >= 0) (void)hipSetDevice(dev);
  }
};

// Rehearsal knob (tests only): MOF_SHARD_SHARE_DEVICE=1 admits the same device for several shards, so the G > 1 slab
// arithmetic (offsets i * slab, ragged and empty last shards) can run on a one-GPU box. Such a group cannot gather (RCCL wants
// one rank per device): gather != 0 is refused.
bool share_device_allowed() {
  const char* v = getenv("MOF_SHARD_SHARE_DEVICE");
  return v && atoi(v) != 0;
}

// what both engine kinds share: the devices, one stream per shard, the communicators
struct ShardCore {
  int n_dev = 0;
  bool shared = false;            // some device carries more than one shard (rehearsal knob)
  std::vector<int> devices;
  std::vector<hipStream_t> streams;
  std::vector<rcclComm_t> comms;  // empty until init_gather (communicators cost ~0.1 s and some device memory)

  int create(const int* devs, int n_devices) {
    if (n_devices < 1 || n_devices > 64) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad shard group (%d devices)", n_devices);
    const int have = mof_device_count();
    if (have <= 0) return mof::capi_fail(MOF_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    n_dev = n_devices;
    for (int i = 0; i < n_devices; ++i) {
      const int d = devs ? devs[i] : i;
      if (d < 0 || d >= have) return mof::capi_fail(MOF_ERR_BAD_ARG, "device %d out of range (0..%d)", d, have - 1);
      for (int j = 0; j < i; ++j)
        if (devices[j] == d) {
          if (!share_device_allowed()) return mof::capi_fail(MOF_ERR_BAD_ARG, "device %d listed twice: one shard per device", d);
          shared = true;
        }
      devices.push_back(d);
      streams.push_back(nullptr);
    }
    for (int i = 0; i < n_devices; ++i) {
      hipError_t he = hipSetDevice(devices[i]);
      if (he == hipSuccess) he = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking);
      if (he != hipSuccess) return mof::capi_fail(MOF_ERR_HIP, "stream on device %d: %s", devices[i], hipGetErrorString(he));
    }
    return MOF_OK;
  }

  // waits for the streams of shards [0, upto): the error path of a batch one of whose shards failed to launch. The failing shard's OWN
  // stream is included by the callers (drain(i + 1)): a batch entry may have enqueued part of its work before it failed (the large-patch
  // pipeline runs several kernels per pass, the half-tile launcher loops over 65535-pair chunks) -- nothing of this call is still running
  // when the caller sees the error, so it may free its buffers
  void drain(int upto) {
    for (int i = 0; i < upto && i < (int)streams.size(); ++i)
      if (streams[i]) {
        (void)hipSetDevice(devices[i]);
        (void)hipStreamSynchronize(streams[i]);
      }
  }

  int sync() {
    for (int i = 0; i < n_dev; ++i) {
      hipError_t he = hipSetDevice(devices[i]);
      if (he == hipSuccess) he = hipStreamSynchronize(streams[i]);
      if (he != hipSuccess) return mof::capi_fail(MOF_ERR_HIP, "sync of shard %d: %s", i, hipGetErrorString(he));
    }
    return MOF_OK;
  }

  // BLOCKING, once: binds RCCL and builds one communicator per device (ncclCommInitAll)
  int init_gather() {
    if (!comms.empty()) return MOF_OK;
    if (shared) return mof::capi_fail(MOF_ERR_UNSUPPORTED, "shards share a device (MOF_SHARD_SHARE_DEVICE rehearsal): RCCL needs one rank per device");
    Rccl* r = rccl();
    if (!r) return mof::capi_fail(MOF_ERR_UNSUPPORTED, "the gather needs RCCL and it cannot be loaded (dlopen of librccl.so.1)");
    comms.assign(n_dev, nullptr);
    const int nc = r->CommInitAll(comms.data(), n_dev, devices.data());
    if (nc != 0) {
      comms.clear();
      return mof::capi_fail(MOF_ERR_HIP, "ncclCommInitAll(%d devices): %s", n_dev, r->GetErrorString(nc));
    }
    return MOF_OK;
  }

  // what RCCL actually formed: ncclCommCount of communicator 0 (0 before init_gather, < 0 on an RCCL error) -- the one fact a
  // reader of a scaling line wants next to the device count
  int gather_ranks() const {
    if (comms.empty()) return 0;
    Rccl* r = rccl();
    int n = 0;
    if (!r || r->CommCount(comms[0], &n) != 0) return mof::capi_fail(MOF_ERR_HIP, "ncclCommCount failed");
    return n;
  }

  // ONE in-place all-gather of `slab_bytes` per rank: rank i's slab already sits at base[i] + i * slab_bytes (the NCCL in-place
  // convention). Asynchronous on the group's streams.
  int gather(unsigned char* const* base, size_t slab_bytes) {
    if (comms.empty()) return mof::capi_fail(MOF_ERR_NOT_INIT, "gather requested before mof_shard_*_init_gather (communicators are not built inside the asynchronous call)");
    Rccl* r = rccl();
    if (!r) return mof::capi_fail(MOF_ERR_UNSUPPORTED, "RCCL is not loaded");
    int nc = r->GroupStart();
    for (int i = 0; i < n_dev && nc == 0; ++i) {
      (void)hipSetDevice(devices[i]);
      nc = r->AllGather(base[i] + (size_t)i * slab_bytes, base[i], slab_bytes, kNcclInt8, comms[i], streams[i]);
    }
    const int ne = r->GroupEnd();
    if (nc != 0 || ne != 0) return mof::capi_fail(MOF_ERR_HIP, "ncclAllGather: %s", r->GetErrorString(nc != 0 ? nc : ne));
    return MOF_OK;
  }

  void destroy_streams_and_comms() {
    Rccl* r = comms.empty() ? nullptr : rccl();
    for (int i = 0; i < n_dev; ++i) {
      if (i >= (int)devices.size()) break;
      (void)hipSetDevice(devices[i]);
      if (i < (int)streams.size() && streams[i]) (void)hipStreamSynchronize(streams[i]);
      if (r && i < (int)comms.size() && comms[i]) (void)r->CommDestroy(comms[i]);
    }
  }
  void destroy_streams() {
    for (int i = 0; i < (int)streams.size(); ++i)
      if (streams[i]) {
        (void)hipSetDevice(devices[i]);
        (void)hipStreamDestroy(streams[i]);
      }
  }
};

}  // namespace

struct mof_shard_fft {
  ShardCore core;
  std::vector<mof_fft_engine*> engines;
  size_t per_pair = 0;  // doubles per frame pair = grid_x * grid_y * 2
};

struct mof_shard_bm {
  ShardCore core;
  std::vector<mof_bm_engine*> engines;
  size_t blocks = 0;  // blocks per frame pair = grid_x * grid_y
};

namespace {
// slab of the block matchers, per rank: [dx: slab_pairs * blocks][dy: slab_pairs * blocks][mode: slab_pairs * 8], rounded up to 16 bytes
size_t bm_slab_bytes(size_t blocks, size_t slab_pairs) { return (slab_pairs * (2 * blocks + 8) + 15) / 16 * 16; }
}  // namespace

extern "C" {

int mof_shard_slab_pairs(int n_pairs, int n_shards) {
  if (n_pairs < 0 || n_shards < 1) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad partition (%d pairs, %d shards)", n_pairs, n_shards);
  return (n_pairs + n_shards - 1) / n_shards;
}

int mof_shard_partition(int n_pairs, int n_shards, int shard, int* first, int* count) {
  if (n_pairs < 0 || n_shards < 1 || shard < 0 || shard >= n_shards || !first || !count)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "bad partition (%d pairs, shard %d of %d)", n_pairs, shard, n_shards);
  const long slab = (n_pairs + n_shards - 1) / n_shards;  // ceil(B / G): SURVEY section 8(e)
  const long lo = slab * shard, hi = lo + slab < n_pairs ? lo + slab : n_pairs;
  *first = (int)(lo < n_pairs ? lo : n_pairs);
  *count = (int)(hi > lo ? hi - lo : 0);
  return MOF_OK;
}

/* ---- FftMethod ---------------------------------------------------------------------------------------------------------- */

void mof_shard_fft_destroy(mof_shard_fft* g) {
  if (!g) return;
  DeviceRestore restore;
  g->core.destroy_streams_and_comms();
  for (size_t i = 0; i < g->engines.size(); ++i)
    if (g->engines[i]) mof_fft_destroy(g->engines[i]);
  g->core.destroy_streams();
  delete g;
}

int mof_shard_fft_create(const mof_fft_config* cfg, const int* devices, int n_devices, mof_shard_fft** out) try {
  if (!out) return mof::capi_fail(MOF_ERR_BAD_ARG, "null out");
  *out = nullptr;
  if (!cfg) return mof::capi_fail(MOF_ERR_BAD_ARG, "null config");
  DeviceRestore restore;
  mof_shard_fft* g = new (std::nothrow) mof_shard_fft();
  if (!g) return mof::capi_fail(MOF_ERR_NO_MEMORY, "out of host memory");
  int rc = g->core.create(devices, n_devices);
  g->per_pair = (size_t)cfg->grid_x * cfg->grid_y * 2;
  for (int i = 0; rc == MOF_OK && i < g->core.n_dev; ++i) {
    mof_fft_config c = *cfg;
    c.device = g->core.devices[i];
    g->engines.push_back(nullptr);
    rc = mof_fft_create(&c, &g->engines[i]);  // (its error text stays the thread's last error)
  }
  if (rc != MOF_OK) {
    mof_shard_fft_destroy(g);
    return rc;
  }
  *out = g;
  return MOF_OK;
} catch (const std::bad_alloc&) {
  return mof::capi_fail(MOF_ERR_NO_MEMORY, "mof_shard_fft_create: out of host memory");
}

int mof_shard_fft_devices(const mof_shard_fft* g) { return g ? g->core.n_dev : 0; }

void* mof_shard_fft_stream(const mof_shard_fft* g, int shard) {
  return (g && shard >= 0 && shard < g->core.n_dev) ? (void*)g->core.streams[shard] : nullptr;
}

int mof_shard_fft_init_gather(mof_shard_fft* g) {
  if (!g) return mof::capi_fail(MOF_ERR_NOT_INIT, "null shard group");
  DeviceRestore restore;
  return g->core.init_gather();
}

int mof_shard_fft_gather_ready(const mof_shard_fft* g) { return (g && !g->core.comms.empty()) ? 1 : 0; }

int mof_shard_fft_gather_ranks(const mof_shard_fft* g) { return g ? g->core.gather_ranks() : 0; }

int mof_shard_fft_process_batch_device(mof_shard_fft* g, const uint8_t* const* d_cur, size_t cur_stride,
                                       const uint8_t* const* d_prev, size_t prev_stride, size_t pitch, int n_pairs,
                                       double* const* d_out, int gather) {
  if (!g) return mof::capi_fail(MOF_ERR_NOT_INIT, "null shard group");
  if (n_pairs == 0) return MOF_OK;
  if (!d_cur || !d_prev || !d_out || n_pairs < 0) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad sharded batch arguments");
  const int G = g->core.n_dev;
  if (gather && g->core.comms.empty())
    return mof::capi_fail(MOF_ERR_NOT_INIT, "gather requested before mof_shard_fft_init_gather (communicators are not built inside the asynchronous call)");
  // arguments first: nothing is launched when any shard's pointers are missing
  for (int i = 0; i < G; ++i) {
    int first = 0, count = 0;
    (void)mof_shard_partition(n_pairs, G, i, &first, &count);
    if (!d_out[i]) return mof::capi_fail(MOF_ERR_BAD_ARG, "null result buffer for shard %d", i);
    if (count > 0 && (!d_cur[i] || !d_prev[i])) return mof::capi_fail(MOF_ERR_BAD_ARG, "null frames for shard %d", i);
  }
  DeviceRestore restore;
  const size_t slab_pairs = (size_t)mof_shard_slab_pairs(n_pairs, G), slab = slab_pairs * g->per_pair;  // doubles per rank
  // every device works on its own shard, on its own stream: the launches return at once and run side by side
  for (int i = 0; i < G; ++i) {
    int first = 0, count = 0;
    (void)mof_shard_partition(n_pairs, G, i, &first, &count);
    if (count == 0) continue;
    const int rc = mof_fft_process_batch_device(g->engines[i], d_cur[i], cur_stride, d_prev[i], prev_stride, pitch, count,
                                                d_out[i] + (size_t)i * slab, g->core.streams[i]);
    if (rc != MOF_OK) {
      g->core.drain(i + 1);  // (the failing call's message stays the thread's last error)
      return rc;
    }
  }
  if (!gather) return MOF_OK;
  std::vector<unsigned char*> base(G);
  for (int i = 0; i < G; ++i) base[i] = reinterpret_cast<unsigned char*>(d_out[i]);
  const int rc = g->core.gather(base.data(), slab * sizeof(double));
  if (rc != MOF_OK) g->core.drain(G);
  return rc;
}

int mof_shard_fft_sync(mof_shard_fft* g) {
  if (!g) return mof::capi_fail(MOF_ERR_NOT_INIT, "null shard group");
  DeviceRestore restore;
  return g->core.sync();
}

/* ---- BlockMethod / FastSpacedBMMethod --------------------------------------------------------------------------------- */

void mof_shard_bm_destroy(mof_shard_bm* g) {
  if (!g) return;
  DeviceRestore restore;
  g->core.destroy_streams_and_comms();
  for (size_t i = 0; i < g->engines.size(); ++i)
    if (g->engines[i]) mof_bm_destroy(g->engines[i]);
  g->core.destroy_streams();
  delete g;
}

int mof_shard_bm_create(const mof_bm_config* cfg, const int* devices, int n_devices, mof_shard_bm** out) try {
  if (!out) return mof::capi_fail(MOF_ERR_BAD_ARG, "null out");
  *out = nullptr;
  if (!cfg) return mof::capi_fail(MOF_ERR_BAD_ARG, "null config");
  DeviceRestore restore;
  mof_shard_bm* g = new (std::nothrow) mof_shard_bm();
  if (!g) return mof::capi_fail(MOF_ERR_NO_MEMORY, "out of host memory");
  int rc = g->core.create(devices, n_devices);
  g->blocks = (size_t)cfg->grid_x * cfg->grid_y;
  for (int i = 0; rc == MOF_OK && i < g->core.n_dev; ++i) {
    mof_bm_config c = *cfg;
    c.device = g->core.devices[i];
    g->engines.push_back(nullptr);
    rc = mof_bm_create(&c, &g->engines[i]);
  }
  if (rc != MOF_OK) {
    mof_shard_bm_destroy(g);
    return rc;
  }
  *out = g;
  return MOF_OK;
} catch (const std::bad_alloc&) {
  return mof::capi_fail(MOF_ERR_NO_MEMORY, "mof_shard_bm_create: out of host memory");
}

int mof_shard_bm_devices(const mof_shard_bm* g) { return g ? g->core.n_dev : 0; }

void* mof_shard_bm_stream(const mof_shard_bm* g, int shard) {
  return (g && shard >= 0 && shard < g->core.n_dev) ? (void*)g->core.streams[shard] : nullptr;
}

int mof_shard_bm_init_gather(mof_shard_bm* g) {
  if (!g) return mof::capi_fail(MOF_ERR_NOT_INIT, "null shard group");
  DeviceRestore restore;
  return g->core.init_gather();
}

int mof_shard_bm_gather_ready(const mof_shard_bm* g) { return (g && !g->core.comms.empty()) ? 1 : 0; }

int mof_shard_bm_gather_ranks(const mof_shard_bm* g) { return g ? g->core.gather_ranks() : 0; }

size_t mof_shard_bm_slab_bytes(const mof_shard_bm* g, int n_pairs) {
  if (!g || n_pairs < 0) return 0;
  return bm_slab_bytes(g->blocks, (size_t)mof_shard_slab_pairs(n_pairs, g->core.n_dev));
}

int mof_shard_bm_locate(const mof_shard_bm* g, int n_pairs, int pair, size_t* dx_off, size_t* dy_off, size_t* mode_off) {
  if (!g) return mof::capi_fail(MOF_ERR_NOT_INIT, "null shard group");
  if (n_pairs < 1 || pair < 0 || pair >= n_pairs) return mof::capi_fail(MOF_ERR_BAD_ARG, "pair %d outside a batch of %d", pair, n_pairs);
  const size_t sp = (size_t)mof_shard_slab_pairs(n_pairs, g->core.n_dev), slab = bm_slab_bytes(g->blocks, sp);
  const size_t shard = (size_t)pair / sp, j = (size_t)pair % sp, base = shard * slab;
  if (dx_off) *dx_off = base + j * g->blocks;
  if (dy_off) *dy_off = base + sp * g->blocks + j * g->blocks;
  if (mode_off) *mode_off = base + 2 * sp * g->blocks + j * 8;
  return MOF_OK;
}

int mof_shard_bm_process_batch_device(mof_shard_bm* g, const uint8_t* const* d_cur, size_t cur_stride,
                                      const uint8_t* const* d_prev, size_t prev_stride, size_t pitch, int n_pairs,
                                      int8_t* const* d_out, int gather) {
  if (!g) return mof::capi_fail(MOF_ERR_NOT_INIT, "null shard group");
  if (n_pairs == 0) return MOF_OK;
  if (!d_cur || !d_prev || !d_out || n_pairs < 0) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad sharded batch arguments");
  const int G = g->core.n_dev;
  if (gather && g->core.comms.empty())
    return mof::capi_fail(MOF_ERR_NOT_INIT, "gather requested before mof_shard_bm_init_gather (communicators are not built inside the asynchronous call)");
  for (int i = 0; i < G; ++i) {
    int first = 0, count = 0;
    (void)mof_shard_partition(n_pairs, G, i, &first, &count);
    if (!d_out[i]) return mof::capi_fail(MOF_ERR_BAD_ARG, "null result buffer for shard %d", i);
    if (count > 0 && (!d_cur[i] || !d_prev[i])) return mof::capi_fail(MOF_ERR_BAD_ARG, "null frames for shard %d", i);
  }
  DeviceRestore restore;
  const size_t sp = (size_t)mof_shard_slab_pairs(n_pairs, G), slab = bm_slab_bytes(g->blocks, sp);
  for (int i = 0; i < G; ++i) {
    int first = 0, count = 0;
    (void)mof_shard_partition(n_pairs, G, i, &first, &count);
    if (count == 0) continue;
    int8_t* s = d_out[i] + (size_t)i * slab;  // this rank's slab, at its place: dx | dy | mode planes
    const int rc = mof_bm_process_batch_device(g->engines[i], d_cur[i], cur_stride, d_prev[i], prev_stride, pitch, count, s,
                                               s + sp * g->blocks, s + 2 * sp * g->blocks, g->core.streams[i]);
    if (rc != MOF_OK) {
      g->core.drain(i + 1);
      return rc;
    }
  }
  if (!gather) return MOF_OK;
  std::vector<unsigned char*> base(G);
  for (int i = 0; i < G; ++i) base[i] = reinterpret_cast<unsigned char*>(d_out[i]);
  const int rc = g->core.gather(base.data(), slab);
  if (rc != MOF_OK) g->core.drain(G);
  return rc;
}

int mof_shard_bm_sync(mof_shard_bm* g) {
  if (!g) return mof::capi_fail(MOF_ERR_NOT_INIT, "null shard group");
  DeviceRestore restore;
  return g->core.sync();
}

}  // extern "C"
