// Cross-replica communication: thin wrappers over RCCL (ncclAllReduce / ncclBroadcast over xGMI),
// one communicator rank per process / GPU. Replaces the reference's cross-clone gradient sum on the
// CPU (slim/deployment/model_deploy.py:414-444 _sum_clones_gradients: tf.add_n of the clones'
// gradients on the optimizer device) and its implicitly shared variables (model_deploy.py:640-675:
// one copy on the CPU, read by every tower) — here every rank holds a replica in HBM, rank 0
// broadcasts the initial values, and the gradient buckets are summed GPU-to-GPU.
//
// RCCL is bound at first use with dlopen (never linked): a process that already carries an RCCL (a
// PyTorch wheel bundles one under the same soname) keeps exactly one copy, a plain C++ host gets
// /opt/rocm's, and the single-GPU entry points of this library load on machines without RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace mtlssl {
namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char path[256] = {0};
};

Rccl g_rccl;
std::once_flag g_once;
char g_load_error[512] = {0};

template <typename F>
bool bind(void* h, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(h, name));
  if (!*out) snprintf(g_load_error, sizeof(g_load_error), "RCCL symbol %s not found: %s", name, dlerror());
  return *out != nullptr;
}

void load_rccl() {
  const char* env = getenv("MTLSSL_RCCL_LIB");
  const char* candidates[] = {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* c : candidates) {
    if (!c || !*c) continue;
    h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (h) { snprintf(g_rccl.path, sizeof(g_rccl.path), "%s", c); break; }
    snprintf(g_load_error, sizeof(g_load_error), "dlopen(%s): %s", c, dlerror());
  }
  if (!h) return;
  Rccl& r = g_rccl;
  bool ok = bind(h, "ncclGetVersion", &r.GetVersion) && bind(h, "ncclGetUniqueId", &r.GetUniqueId) &&
            bind(h, "ncclCommInitRank", &r.CommInitRank) && bind(h, "ncclCommDestroy", &r.CommDestroy) &&
            bind(h, "ncclCommCount", &r.CommCount) && bind(h, "ncclCommUserRank", &r.CommUserRank) &&
            bind(h, "ncclCommCuDevice", &r.CommCuDevice) && bind(h, "ncclAllReduce", &r.AllReduce) &&
            bind(h, "ncclBroadcast", &r.Broadcast) && bind(h, "ncclGetErrorString", &r.GetErrorString);
  if (ok) r.handle = h;
}

const Rccl* rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.handle) {
    set_error("comm: RCCL is not available (%s)", g_load_error);
    return nullptr;
  }
  return &g_rccl;
}

int nccl_fail(const Rccl* r, const char* what, ncclResult_t e) {
  set_error("comm: %s: %s", what, r->GetErrorString(e));
  return MTLSSL_ECOMM;
}

bool dtype_of(int dtype, ncclDataType_t* t, size_t* size) {
  switch (dtype) {
    case MTLSSL_COMM_F32: *t = ncclFloat32; *size = 4; return true;
    case MTLSSL_COMM_F64: *t = ncclFloat64; *size = 8; return true;
    case MTLSSL_COMM_I32: *t = ncclInt32; *size = 4; return true;
    case MTLSSL_COMM_I64: *t = ncclInt64; *size = 8; return true;
  }
  return false;
}

}  // namespace
}  // namespace mtlssl

struct mtlssl_comm {
  ncclComm_t nccl;
  int nranks, rank, device;
};

using namespace mtlssl;

extern "C" {

int mtlssl_comm_available(int* rccl_version_out, int* device_out) {
  const Rccl* r = rccl();
  if (!r) return MTLSSL_ECOMM;
  int v = 0;
  ncclResult_t e = r->GetVersion(&v);
  if (e != ncclSuccess) return nccl_fail(r, "ncclGetVersion", e);
  int dev = -1;
  hipError_t he = hipGetDevice(&dev);
  if (he != hipSuccess || dev < 0) {
    set_error("comm: no usable HIP device is current (%s)", hipGetErrorString(he));
    return MTLSSL_ECOMM;
  }
  if (rccl_version_out) *rccl_version_out = v;
  if (device_out) *device_out = dev;
  return MTLSSL_OK;
}

int mtlssl_comm_unique_id(void* id_out) {
  MTLSSL_REQUIRE(id_out != nullptr, "comm_unique_id: null output");
  const Rccl* r = rccl();
  if (!r) return MTLSSL_ECOMM;
  static_assert(sizeof(ncclUniqueId) == MTLSSL_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId", e);
  memcpy(id_out, &id, sizeof(id));
  return MTLSSL_OK;
}

int mtlssl_comm_init(mtlssl_comm_t* comm_out, const void* id, int nranks, int rank) {
  MTLSSL_REQUIRE(comm_out != nullptr && id != nullptr, "comm_init: null argument");
  MTLSSL_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: rank %d of %d", rank, nranks);
  const Rccl* r = rccl();
  if (!r) return MTLSSL_ECOMM;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  mtlssl_comm* c = new mtlssl_comm();
  ncclResult_t e = r->CommInitRank(&c->nccl, nranks, uid, rank);   // binds to the calling thread's current device
  if (e != ncclSuccess) { delete c; return nccl_fail(r, "ncclCommInitRank", e); }
  // what RCCL itself reports — not what the caller asked for
  if ((e = r->CommCount(c->nccl, &c->nranks)) != ncclSuccess || (e = r->CommUserRank(c->nccl, &c->rank)) != ncclSuccess ||
      (e = r->CommCuDevice(c->nccl, &c->device)) != ncclSuccess) {
    r->CommDestroy(c->nccl);
    delete c;
    return nccl_fail(r, "ncclComm{Count,UserRank,CuDevice}", e);
  }
  *comm_out = c;
  return MTLSSL_OK;
}

int mtlssl_comm_info(mtlssl_comm_t comm, int* nranks, int* rank, int* device, int* rccl_version) {
  MTLSSL_REQUIRE(comm != nullptr, "comm_info: null communicator");
  const Rccl* r = rccl();
  if (!r) return MTLSSL_ECOMM;
  if (nranks) *nranks = comm->nranks;
  if (rank) *rank = comm->rank;
  if (device) *device = comm->device;
  if (rccl_version) {
    ncclResult_t e = r->GetVersion(rccl_version);
    if (e != ncclSuccess) return nccl_fail(r, "ncclGetVersion", e);
  }
  return MTLSSL_OK;
}

int mtlssl_comm_allreduce(mtlssl_comm_t comm, void* buf, int64_t count, int dtype, int op, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(comm != nullptr, "comm_allreduce: null communicator");
  MTLSSL_REQUIRE(count >= 0 && (buf != nullptr || count == 0), "comm_allreduce: bad buffer");
  ncclDataType_t t; size_t sz;
  MTLSSL_REQUIRE(dtype_of(dtype, &t, &sz), "comm_allreduce: dtype %d", dtype);
  MTLSSL_REQUIRE(op >= MTLSSL_COMM_SUM && op <= MTLSSL_COMM_MIN, "comm_allreduce: op %d", op);
  if (count == 0) return MTLSSL_OK;
  const Rccl* r = rccl();
  if (!r) return MTLSSL_ECOMM;
  const ncclRedOp_t ops[] = {ncclSum, ncclMax, ncclMin};
  ncclResult_t e = r->AllReduce(buf, buf, (size_t)count, t, ops[op], comm->nccl, S(stream));   // in place
  if (e != ncclSuccess) return nccl_fail(r, "ncclAllReduce", e);
  return MTLSSL_OK;
}

int mtlssl_comm_broadcast(mtlssl_comm_t comm, void* buf, int64_t bytes, int root, mtlssl_stream_t stream) {
  MTLSSL_REQUIRE(comm != nullptr, "comm_broadcast: null communicator");
  MTLSSL_REQUIRE(bytes >= 0 && (buf != nullptr || bytes == 0), "comm_broadcast: bad buffer");
  MTLSSL_REQUIRE(root >= 0 && root < comm->nranks, "comm_broadcast: root %d of %d", root, comm->nranks);
  if (bytes == 0) return MTLSSL_OK;
  const Rccl* r = rccl();
  if (!r) return MTLSSL_ECOMM;
  ncclResult_t e = r->Broadcast(buf, buf, (size_t)bytes, ncclInt8, root, comm->nccl, S(stream));
  if (e != ncclSuccess) return nccl_fail(r, "ncclBroadcast", e);
  return MTLSSL_OK;
}

// CRC-32C (Castagnoli, reflected 0x82F63B78), slicing-by-8 on the host.
int64_t mtlssl_crc32c_host(const void* data, int64_t nbytes, int64_t crc_in) {
  static uint32_t T[8][256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
      T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
  });
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t crc = ~(uint32_t)crc_in;
  while (nbytes >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
    lo ^= crc;
    crc = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^
          T[3][hi & 0xFF] ^ T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
    p += 8; nbytes -= 8;
  }
  while (nbytes-- > 0) crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return (int64_t)(uint32_t)~crc;
}

int mtlssl_comm_destroy(mtlssl_comm_t comm) {
  if (!comm) return MTLSSL_OK;
  const Rccl* r = rccl();
  if (!r) return MTLSSL_ECOMM;
  ncclResult_t e = r->CommDestroy(comm->nccl);
  delete comm;
  if (e != ncclSuccess) return nccl_fail(r, "ncclCommDestroy", e);
  return MTLSSL_OK;
}

}  // extern "C"
