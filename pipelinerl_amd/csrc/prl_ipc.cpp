// prl_ipc: weight hand-off between processes that share ONE GPU (BASELINE config 2: actor and learner
// colocated on a single MI355X).  There is nothing to send over xGMI in that case: the trainer
// flattens its parameters into a device bucket allocated here, exports a HIP IPC handle, and the
// inference worker maps the same HBM pages and copies them into its own weights (device-to-device,
// ~HBM copy speed).  The reference has no special colocated path: it still runs its per-parameter
// NCCL broadcast between the two processes (finetune_loop.py:279-282, vllm1.py:118-122).
//
// Buckets are plain hipMalloc allocations (IPC handles refer to whole allocations; PyTorch's caching
// allocator sub-allocates, so its pointers cannot be exported directly).
#include <hip/hip_runtime_api.h>

#include <cstring>

#include "prl_common.h"

static_assert(sizeof(hipIpcMemHandle_t) == PRL_IPC_HANDLE_BYTES, "IPC handle size");

#define PRL_IPC_CHECK(expr)                                                                        \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return prl::set_error(PRL_EFAULT, "%s failed: %s", #expr, hipGetErrorString(_e));            \
  } while (0)

extern "C" int prl_ipc_alloc(uint64_t nbytes, void** dev_ptr) {
  PRL_CHECK_ARG(dev_ptr != nullptr && nbytes > 0, "bad argument");
  PRL_IPC_CHECK(hipMalloc(dev_ptr, (size_t)nbytes));
  return PRL_OK;
}

extern "C" int prl_ipc_free(void* dev_ptr) {
  if (!dev_ptr) return PRL_OK;
  PRL_IPC_CHECK(hipFree(dev_ptr));
  return PRL_OK;
}

extern "C" int prl_ipc_export(const void* dev_ptr, uint8_t handle[PRL_IPC_HANDLE_BYTES]) {
  PRL_CHECK_ARG(dev_ptr != nullptr && handle != nullptr, "null argument");
  hipIpcMemHandle_t h;
  PRL_IPC_CHECK(hipIpcGetMemHandle(&h, const_cast<void*>(dev_ptr)));
  memcpy(handle, &h, sizeof(h));
  return PRL_OK;
}

extern "C" int prl_ipc_open(const uint8_t handle[PRL_IPC_HANDLE_BYTES], void** dev_ptr) {
  PRL_CHECK_ARG(dev_ptr != nullptr && handle != nullptr, "null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  PRL_IPC_CHECK(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
  return PRL_OK;
}

extern "C" int prl_ipc_close(void* dev_ptr) {
  if (!dev_ptr) return PRL_OK;
  PRL_IPC_CHECK(hipIpcCloseMemHandle(dev_ptr));
  return PRL_OK;
}
