// Gradient-bucket all-reduce over RCCL (xGMI), behind the C ABI.
//
// The reference has no distributed code; this is the collective of the build's
// data-parallel row (SURVEY.md 8e / 8b `lss_allreduce_bucket`): one in-place fp32 sum over a slice
// of the flat gradient buffer, enqueued on the caller's stream so that it is ordered behind the
// backward kernels that produced the slice (lss2_multimodal_nu_amd/dp.py starts it from a
// backward hook).
//
// No link-time dependency on RCCL: the symbols are resolved at first use from the librccl that
// is ALREADY loaded in the process (a PyTorch-ROCm process carries its own copy in torch/lib; a
// second copy of the library next to it would double the symbol set), falling back to the
// system's librccl.so for a plain C caller.  The header is used for the types only.
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "lss_common.h"

namespace {

struct RcclApi {
  ncclResult_t (*GetVersion)(int*);
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  const char* (*GetErrorString)(ncclResult_t);
  bool ok;
};

RcclApi g_api;
std::once_flag g_once;
char g_path[1024];

int find_loaded(struct dl_phdr_info* info, size_t, void*) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
    strncpy(g_path, info->dlpi_name, sizeof(g_path) - 1);
    return 1;
  }
  return 0;
}

void resolve() {
  g_api.ok = false;
  g_path[0] = 0;
  dl_iterate_phdr(find_loaded, nullptr);
  void* h = g_path[0] ? dlopen(g_path, RTLD_NOW | RTLD_NOLOAD) : nullptr;
  if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) return;
  g_api.GetVersion = reinterpret_cast<decltype(g_api.GetVersion)>(dlsym(h, "ncclGetVersion"));
  g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_api.AllReduce = reinterpret_cast<decltype(g_api.AllReduce)>(dlsym(h, "ncclAllReduce"));
  g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  g_api.ok = g_api.GetVersion && g_api.GetUniqueId && g_api.CommInitRank && g_api.AllReduce && g_api.CommDestroy;
}

const RcclApi* api() {
  std::call_once(g_once, resolve);
  return g_api.ok ? &g_api : nullptr;
}

// ncclResult_t r -> ABI code: 0, or LSS_E_RCCL_BASE - r (a distinct negative range)
inline int rc(ncclResult_t r) { return r == ncclSuccess ? 0 : LSS_E_RCCL_BASE - (int)r; }

}  // namespace

extern "C" size_t lss_rccl_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

extern "C" int lss_rccl_version(int* version) {
  LSS_CHECK_PTR(version);
  const RcclApi* a = api();
  if (a == nullptr) return LSS_E_RCCL_BASE;
  return rc(a->GetVersion(version));
}

extern "C" int lss_rccl_get_unique_id(void* id_host) {
  LSS_CHECK_PTR(id_host);
  const RcclApi* a = api();
  if (a == nullptr) return LSS_E_RCCL_BASE;
  ncclUniqueId id;
  const ncclResult_t r = a->GetUniqueId(&id);
  if (r == ncclSuccess) memcpy(id_host, &id, sizeof(id));
  return rc(r);
}

extern "C" int lss_rccl_comm_init(const void* id_host, int nranks, int rank, void** comm) {
  LSS_CHECK_PTR(id_host); LSS_CHECK_PTR(comm);
  if (nranks <= 0 || rank < 0 || rank >= nranks) return LSS_E_SHAPE;
  const RcclApi* a = api();
  if (a == nullptr) return LSS_E_RCCL_BASE;
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = a->CommInitRank(&c, nranks, id, rank);
  *comm = r == ncclSuccess ? reinterpret_cast<void*>(c) : nullptr;
  return rc(r);
}

extern "C" int lss_allreduce_bucket(void* comm, float* buf, long long n, void* stream) {
  LSS_CHECK_PTR(comm); LSS_CHECK_PTR(buf);
  if (n <= 0) return LSS_E_SHAPE;
  const RcclApi* a = api();
  if (a == nullptr) return LSS_E_RCCL_BASE;
  return rc(a->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                         lss_stream(stream)));
}

extern "C" int lss_rccl_comm_destroy(void* comm) {
  LSS_CHECK_PTR(comm);
  const RcclApi* a = api();
  if (a == nullptr) return LSS_E_RCCL_BASE;
  return rc(a->CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
}
