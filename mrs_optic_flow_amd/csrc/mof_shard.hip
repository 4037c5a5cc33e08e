// mof_shard.hip -- the batched-frames mode across the GPUs of one node, natively behind the C ABI (mof_shard_* in include/mof.h).
//
// BASELINE north_star / SURVEY section 8(e): frame pairs are independent, so a batch of B pairs is cut into contiguous shards of
// ceil(B / G) pairs, shard g lives on (and is processed by) device g -- no data-path collective -- and ONE all-gather of the
// per-rank result slabs (RCCL over xGMI; ceil(B/G) * grid * 2 doubles per rank) hands every device the whole result. The host
// side stays C++ (the reference is a C++ ROS node) and needs no Python: ONE process, one FftMethod engine + one HIP stream per
// device, ncclCommInitAll for the communicators, ncclGroupStart / ncclAllGather (in place) / ncclGroupEnd per batch.
// RCCL is bound at run time (dlopen of librccl.so.1 on first use): a single-GPU host -- the ROS node -- never maps the 570 MB
// library, and a process that already holds a copy (PyTorch bundles one under the same SONAME) shares it.
// The reference has no counterpart (one synchronous call per frame on one device); the per-device work is exactly
// mof_fft_process_batch_device / mof_bm_process_batch_device.

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
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
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return (r.handle && !r.why[0]) ? &r : nullptr;
}

}  // namespace

struct mof_shard_fft {
  int n_dev = 0;
  std::vector<int> devices;
  std::vector<mof_fft_engine*> engines;
  std::vector<hipStream_t> streams;
  std::vector<rcclComm_t> comms;  // empty until the first gather (communicators cost ~0.1 s and some device memory)
  size_t per_pair = 0;            // doubles per frame pair = grid_x * grid_y * 2
};

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

void mof_shard_fft_destroy(mof_shard_fft* g) {
  if (!g) return;
  Rccl* r = g->comms.empty() ? nullptr : rccl();
  for (int i = 0; i < g->n_dev; ++i) {
    (void)hipSetDevice(g->devices[i]);
    if (i < (int)g->streams.size() && g->streams[i]) (void)hipStreamSynchronize(g->streams[i]);
    if (r && i < (int)g->comms.size() && g->comms[i]) (void)r->CommDestroy(g->comms[i]);
    if (i < (int)g->engines.size() && g->engines[i]) mof_fft_destroy(g->engines[i]);
    if (i < (int)g->streams.size() && g->streams[i]) (void)hipStreamDestroy(g->streams[i]);
  }
  delete g;
}

int mof_shard_fft_create(const mof_fft_config* cfg, const int* devices, int n_devices, mof_shard_fft** out) try {
  if (!out) return mof::capi_fail(MOF_ERR_BAD_ARG, "null out");
  *out = nullptr;
  if (!cfg || n_devices < 1 || n_devices > 64) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad shard group (%d devices)", n_devices);
  const int have = mof_device_count();
  if (have <= 0) return mof::capi_fail(MOF_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  mof_shard_fft* g = new (std::nothrow) mof_shard_fft();
  if (!g) return mof::capi_fail(MOF_ERR_NO_MEMORY, "out of host memory");
  g->n_dev = n_devices;
  g->per_pair = (size_t)cfg->grid_x * cfg->grid_y * 2;
  for (int i = 0; i < n_devices; ++i) {
    const int d = devices ? devices[i] : i;
    for (int j = 0; j < i; ++j)
      if (g->devices[j] == d) {
        mof_shard_fft_destroy(g);
        return mof::capi_fail(MOF_ERR_BAD_ARG, "device %d listed twice: one shard per device", d);
      }
    if (d < 0 || d >= have) {
      mof_shard_fft_destroy(g);
      return mof::capi_fail(MOF_ERR_BAD_ARG, "device %d out of range (0..%d)", d, have - 1);
    }
    g->devices.push_back(d);
    g->engines.push_back(nullptr);
    g->streams.push_back(nullptr);
  }
  for (int i = 0; i < n_devices; ++i) {
    mof_fft_config c = *cfg;
    c.device = g->devices[i];
    const int rc = mof_fft_create(&c, &g->engines[i]);  // (its error text stays the thread's last error)
    if (rc != MOF_OK) {
      mof_shard_fft_destroy(g);
      return rc;
    }
    hipError_t he = hipSetDevice(g->devices[i]);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&g->streams[i], hipStreamNonBlocking);
    if (he != hipSuccess) {
      mof_shard_fft_destroy(g);
      return mof::capi_fail(MOF_ERR_HIP, "stream on device %d: %s", g->devices[i], hipGetErrorString(he));
    }
  }
  *out = g;
  return MOF_OK;
} catch (const std::bad_alloc&) {
  return mof::capi_fail(MOF_ERR_NO_MEMORY, "mof_shard_fft_create: out of host memory");
}

int mof_shard_fft_devices(const mof_shard_fft* g) { return g ? g->n_dev : 0; }

void* mof_shard_fft_stream(const mof_shard_fft* g, int shard) {
  return (g && shard >= 0 && shard < g->n_dev) ? (void*)g->streams[shard] : nullptr;
}

int mof_shard_fft_process_batch_device(mof_shard_fft* g, const uint8_t* const* d_cur, size_t cur_stride,
                                       const uint8_t* const* d_prev, size_t prev_stride, size_t pitch, int n_pairs,
                                       double* const* d_out, int gather) {
  if (!g) return mof::capi_fail(MOF_ERR_NOT_INIT, "null shard group");
  if (n_pairs == 0) return MOF_OK;
  if (!d_cur || !d_prev || !d_out || n_pairs < 0) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad sharded batch arguments");
  const int G = g->n_dev;
  const size_t slab_pairs = (size_t)mof_shard_slab_pairs(n_pairs, G), slab = slab_pairs * g->per_pair;  // doubles per rank
  // every device works on its own shard, on its own stream: the launches return at once and run side by side
  for (int i = 0; i < G; ++i) {
    int first = 0, count = 0;
    (void)mof_shard_partition(n_pairs, G, i, &first, &count);
    if (!d_out[i]) return mof::capi_fail(MOF_ERR_BAD_ARG, "null result buffer for shard %d", i);
    if (count == 0) continue;
    if (!d_cur[i] || !d_prev[i]) return mof::capi_fail(MOF_ERR_BAD_ARG, "null frames for shard %d", i);
    const int rc = mof_fft_process_batch_device(g->engines[i], d_cur[i], cur_stride, d_prev[i], prev_stride, pitch, count,
                                                d_out[i] + (size_t)i * slab, g->streams[i]);
    if (rc != MOF_OK) return rc;
  }
  if (!gather || G == 0) return MOF_OK;
  Rccl* r = rccl();
  if (!r) return mof::capi_fail(MOF_ERR_UNSUPPORTED, "the gather needs RCCL and it cannot be loaded: %s", rccl() ? "" : "dlopen(librccl.so.1) failed");
  if (g->comms.empty()) {
    g->comms.assign(G, nullptr);
    const int nc = r->CommInitAll(g->comms.data(), G, g->devices.data());
    if (nc != 0) {
      g->comms.clear();
      return mof::capi_fail(MOF_ERR_HIP, "ncclCommInitAll(%d devices): %s", G, r->GetErrorString(nc));
    }
  }
  // ONE in-place all-gather: rank i's slab already sits at d_out[i] + i * slab (the NCCL in-place convention)
  int nc = r->GroupStart();
  for (int i = 0; i < G && nc == 0; ++i) {
    (void)hipSetDevice(g->devices[i]);
    nc = r->AllGather(d_out[i] + (size_t)i * slab, d_out[i], slab * sizeof(double), kNcclInt8, g->comms[i], g->streams[i]);
  }
  const int ne = r->GroupEnd();
  if (nc != 0 || ne != 0) return mof::capi_fail(MOF_ERR_HIP, "ncclAllGather: %s", r->GetErrorString(nc != 0 ? nc : ne));
  return MOF_OK;
}

int mof_shard_fft_sync(mof_shard_fft* g) {
  if (!g) return mof::capi_fail(MOF_ERR_NOT_INIT, "null shard group");
  for (int i = 0; i < g->n_dev; ++i) {
    hipError_t he = hipSetDevice(g->devices[i]);
    if (he == hipSuccess) he = hipStreamSynchronize(g->streams[i]);
    if (he != hipSuccess) return mof::capi_fail(MOF_ERR_HIP, "sync of shard %d: %s", i, hipGetErrorString(he));
  }
  return MOF_OK;
}

}  // extern "C"
