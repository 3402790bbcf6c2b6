// Multi-GPU film reassembly behind the C ABI (include/rrt.h): RCCL over xGMI, one collective per frame.
// The reference has one address space and merges tiles under a lock (Film::merge_film_tile film.rs:248-263, driven from
// integrator/mod.rs:64-74,133); here every rank renders the interleaved 16-row bands b % world == rank of the frame into
// its own device film (rrt_render_bands) and the films meet on `root`:
//   box filter, radius <= 0.5  every sample lands in its own pixel, the ranks' bands are disjoint: grouped ncclSend /
//                              ncclRecv of the band rows only - a gather; each rank ships 1/world of the film over its
//                              direct xGMI link to root (root receives from all peers at once: no ring);
//   wider filters              samples splat into neighbouring rows of the rank's own film: ncclReduce(sum) of the film.
// A band is contiguous in the film (rows x W x 4 words), so no packing is needed.
// RCCL is bound at the first collective call (dlopen of the librccl.so.1 the process already holds - torch.distributed's, for instance - or
// of the ROCm one), not linked into librrt.so: a single-GPU user never loads it, and a process that loads librrt.so before PyTorch does not
// end up with two copies of RCCL's SMI library whose static destructors then free the same tables twice at exit.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>

#include "rrt_impl.hpp"

namespace rrt { void set_last_error(const std::string& msg); }
struct rrt_handle { rrtd::HandleBase* impl; };
struct rrt_comm { ncclComm_t comm; int rank, world, device; };

namespace {
constexpr int kBandRows = 16;   // tile height of integrator/mod.rs:55 (rrt_render_bands)

struct NcclError : std::runtime_error { using std::runtime_error::runtime_error; };

// the RCCL entry points this file calls, resolved once
struct Rccl {
  decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&::ncclCommInitRank) CommInitRank = nullptr;
  decltype(&::ncclCommInitAll) CommInitAll = nullptr;
  decltype(&::ncclCommDestroy) CommDestroy = nullptr;
  decltype(&::ncclCommAbort) CommAbort = nullptr;
  decltype(&::ncclGroupStart) GroupStart = nullptr;
  decltype(&::ncclGroupEnd) GroupEnd = nullptr;
  decltype(&::ncclSend) Send = nullptr;
  decltype(&::ncclRecv) Recv = nullptr;
  decltype(&::ncclReduce) Reduce = nullptr;
  decltype(&::ncclGetErrorString) GetErrorString = nullptr;
};
const Rccl& rccl() {
  static Rccl api;
  static std::once_flag once;
  static std::string failure;
  std::call_once(once, []() {
    void* lib = nullptr;
    std::string tried;
    // RRT_RCCL_LIBRARY: the one library to bind (a deployment that ships its own RCCL; tests point it at a missing file)
    const char* forced = getenv("RRT_RCCL_LIBRARY");
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      const char* path = (forced && *forced) ? forced : name;
      if ((lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL))) break;
      const char* e = dlerror();   // (one call: dlerror() clears the message it returns)
      if (tried.empty()) tried = e ? e : "?";
      if (forced && *forced) break;
    }
    if (!lib) { failure = "RCCL not found (dlopen librccl.so.1): " + tried; return; }
    auto sym = [&](const char* n) { void* f = dlsym(lib, n); if (!f && failure.empty()) failure = std::string("RCCL symbol missing: ") + n; return f; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.Reduce = (decltype(api.Reduce))sym("ncclReduce");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  });
  if (!failure.empty()) throw NcclError(failure);
  return api;
}
#define NCCL_CHECK(expr)                                                                                                \
  do {                                                                                                                  \
    ncclResult_t _r = (expr);                                                                                           \
    if (_r != ncclSuccess) throw NcclError(std::string("RCCL error: ") + rccl().GetErrorString(_r) + " at " #expr);    \
  } while (0)

template <typename F>
int guarded(F&& fn) {
  try { fn(); return RRT_OK; }
  catch (const NcclError& e) { rrt::set_last_error(e.what()); return RRT_EDEVICE; }
  catch (const rrtd::DeviceError& e) { rrt::set_last_error(e.what()); return RRT_EDEVICE; }
  catch (const std::invalid_argument& e) { rrt::set_last_error(e.what()); return RRT_EINVAL; }
  catch (const std::exception& e) { rrt::set_last_error(e.what()); return RRT_EINVAL; }
}

// communicators of rrt_film_gather_all (one process, all GPUs): created on first use, kept while handles on those devices live
// (rrt::comm_cache_release, called from rrt_destroy), keyed by the device list
std::mutex g_cache_mu;
std::vector<int> g_cache_devs;
std::vector<ncclComm_t> g_cache_comms;
void cache_drop_locked(bool abort) {
  for (ncclComm_t c : g_cache_comms) if (c) { if (abort) (void)rccl().CommAbort(c); else (void)rccl().CommDestroy(c); }
  g_cache_comms.clear(); g_cache_devs.clear();
}

// One rank's part of the collective, in two steps so that nothing that can throw for a host-side reason runs inside an open ncclGroup:
// plan_gather() reads the film geometry and lists the transfers (before ncclGroupStart), enqueue_gather() issues them (inside the group).
struct GatherOp { int kind; void* at; size_t count; int peer; };   // kind 0 = reduce (in place on root), 1 = recv, 2 = send
struct GatherPlan { std::vector<GatherOp> ops; ncclDataType_t dt = ncclFloat; hipStream_t st = nullptr; };
GatherPlan plan_gather(rrtd::HandleBase* h, int rank, int world, void* film, int root) {
  GatherPlan p;
  int W = 0, H = 0;
  bool splats = false;
  h->film_geometry(&W, &H, &splats);
  if (W < 1 || H < 1) throw std::invalid_argument("rrt_film_gather: empty film");
  p.dt = h->precision() == RRT_F32 ? ncclFloat : ncclDouble;
  const size_t word = h->precision() == RRT_F32 ? 4 : 8;
  p.st = h->stream();
  if (splats) {   // overlapping films: sum (in place on root)
    p.ops.push_back(GatherOp{0, film, (size_t)W * (size_t)H * 4, root});
    return p;
  }
  const int n_bands = (H + kBandRows - 1) / kBandRows;
  for (int b = 0; b < n_bands; b++) {
    const int owner = b % world;
    if (owner == root) continue;                 // root's own bands are already in place
    if (rank != root && rank != owner) continue;
    const int y0 = b * kBandRows, y1 = std::min(H, y0 + kBandRows);
    char* at = (char*)film + (size_t)y0 * (size_t)W * 4 * word;
    p.ops.push_back(GatherOp{rank == root ? 1 : 2, at, (size_t)(y1 - y0) * (size_t)W * 4, rank == root ? owner : root});
  }
  return p;
}
// RRT_TEST_FAIL_GATHER=<rank>: that rank's first transfer is issued with an invalid peer, so that the enqueue fails INSIDE the group
// (tests/test_gpu_multi.py: the error path below must close the group before it aborts the communicator)
void enqueue_gather(const GatherPlan& p, ncclComm_t comm, int rank, int world, int root) {
  const char* tf = getenv("RRT_TEST_FAIL_GATHER");
  const bool inject = tf && *tf && atoi(tf) == rank;
  if (inject && p.ops.empty()) throw NcclError("RCCL error: injected failure (RRT_TEST_FAIL_GATHER) on a rank with nothing to transfer");
  bool first = true;
  for (const GatherOp& op : p.ops) {
    const int peer = (inject && first) ? world + 7 : op.peer;
    first = false;
    if (op.kind == 0) NCCL_CHECK(rccl().Reduce(op.at, op.at, op.count, p.dt, ncclSum, inject ? world + 7 : root, comm, p.st));
    else if (op.kind == 1) NCCL_CHECK(rccl().Recv(op.at, op.count, p.dt, peer, comm, p.st));
    else NCCL_CHECK(rccl().Send(op.at, op.count, p.dt, peer, comm, p.st));
  }
}
}  // namespace

extern "C" {

int rrt_band_rows(int yres, int rank, int world, int32_t* y0y1, int max_bands) {
  if (yres < 0 || world < 1 || rank < 0 || rank >= world) { rrt::set_last_error("rrt_band_rows: bad argument"); return RRT_EINVAL; }
  int n = 0;
  for (int b = 0, y0 = 0; y0 < yres; b++, y0 += kBandRows) {
    if (b % world != rank) continue;
    if (y0y1 && n < max_bands) { y0y1[2 * n] = y0; y0y1[2 * n + 1] = std::min(yres, y0 + kBandRows); }
    n++;
  }
  return n;
}

int rrt_comm_id(uint8_t id[RRT_COMM_ID_BYTES]) {
  static_assert(RRT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "rrt.h mirrors ncclUniqueId");
  if (!id) { rrt::set_last_error("rrt_comm_id: null argument"); return RRT_EINVAL; }
  return guarded([&]() {
    ncclUniqueId u;
    NCCL_CHECK(rccl().GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  });
}

int rrt_comm_create(const uint8_t id[RRT_COMM_ID_BYTES], int rank, int world, int device, rrt_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) { rrt::set_last_error("rrt_comm_create: bad argument"); return RRT_EINVAL; }
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); rrt::set_last_error("rrt_comm_create: no HIP device visible"); return RRT_EDEVICE; }
  if (device < 0 || device >= n) { rrt::set_last_error("rrt_comm_create: device index out of range"); return RRT_EINVAL; }
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    NCCL_CHECK(rccl().CommInitRank(&c, world, u, rank));
    *out = new rrt_comm{c, rank, world, device};
  });
}

void rrt_comm_destroy(rrt_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm) { try { (void)rccl().CommDestroy(c->comm); } catch (const NcclError&) {} }   // (null: aborted after a failed collective)
  delete c;
}

int rrt_film_gather(rrt_handle* h, rrt_comm* c, void* film_xyzw_device, int root) {
  if (!h || !c || !film_xyzw_device) { rrt::set_last_error("rrt_film_gather: null argument"); return RRT_EINVAL; }
  if (root < 0 || root >= c->world) { rrt::set_last_error("rrt_film_gather: root out of range"); return RRT_EINVAL; }
  if (h->impl->device() != c->device) { rrt::set_last_error("rrt_film_gather: handle and communicator live on different devices"); return RRT_EINVAL; }
  if (!c->comm) { rrt::set_last_error("rrt_film_gather: the communicator was aborted by an earlier failed collective"); return RRT_EDEVICE; }
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(c->device));
    const GatherPlan plan = plan_gather(h->impl, c->rank, c->world, film_xyzw_device, root);   // everything that can throw for host-side reasons: before the group
    h->impl->gather_mark(true);
    NCCL_CHECK(rccl().GroupStart());
    // A failure inside the group leaves it partly enqueued. The order of the clean-up matters: ncclGroupEnd FIRST - RCCL discards a group that
    // holds a failed call and returns the error without launching anything, and it must walk this thread's group state while the communicator
    // is still alive - and only then ncclCommAbort (which frees the communicator: the peers' calls fail instead of waiting for transfers that
    // will never be matched), after which the rrt_comm refuses further collectives.
    try { enqueue_gather(plan, c->comm, c->rank, c->world, root); }
    catch (...) { (void)rccl().GroupEnd(); (void)rccl().CommAbort(c->comm); c->comm = nullptr; throw; }
    NCCL_CHECK(rccl().GroupEnd());
    h->impl->gather_mark(false);
  });
}

// One process that owns all GPUs of the node (what the reference's single binary becomes): communicators come from
// ncclCommInitAll over the handles' devices and are kept until a handle on one of those devices is destroyed (keyed by the device
// list). n == 1: the frame is already where it belongs, nothing is sent and RCCL is not even loaded.
int rrt_film_gather_all(rrt_handle* const* handles, void* const* films_device, int n, int root) {
  if (!handles || !films_device || n < 1 || root < 0 || root >= n) { rrt::set_last_error("rrt_film_gather_all: bad argument"); return RRT_EINVAL; }
  for (int i = 0; i < n; i++) if (!handles[i] || !films_device[i]) { rrt::set_last_error("rrt_film_gather_all: null handle / film"); return RRT_EINVAL; }
  if (n == 1) return RRT_OK;
  return guarded([&]() {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    std::vector<int> want(n);
    for (int i = 0; i < n; i++) want[i] = handles[i]->impl->device();
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (want[i] == want[j]) throw std::invalid_argument("rrt_film_gather_all: two handles on one device (RCCL needs one rank per GPU)");
    if (want != g_cache_devs) {
      cache_drop_locked(false);
      g_cache_comms.assign(n, nullptr);
      NCCL_CHECK(rccl().CommInitAll(g_cache_comms.data(), n, want.data()));
      g_cache_devs = want;
    }
    std::vector<GatherPlan> plans;
    for (int i = 0; i < n; i++) plans.push_back(plan_gather(handles[i]->impl, i, n, films_device[i], root));
    for (int i = 0; i < n; i++) handles[i]->impl->gather_mark(true);
    NCCL_CHECK(rccl().GroupStart());
    try {
      for (int i = 0; i < n; i++) {
        HIP_CHECK(hipSetDevice(want[i]));
        enqueue_gather(plans[i], g_cache_comms[i], i, n, root);
      }
    } catch (...) { (void)rccl().GroupEnd(); cache_drop_locked(true); throw; }   // (see rrt_film_gather: close the group first, then abort the communicators)
    NCCL_CHECK(rccl().GroupEnd());
    for (int i = 0; i < n; i++) handles[i]->impl->gather_mark(false);
  });
}

}  // extern "C"

// rrt_destroy (rrt_api.hip): a handle on `device` goes away - so do the cached communicators that span that device
namespace rrt {
void comm_cache_release(int device) {
  std::lock_guard<std::mutex> lock(g_cache_mu);
  if (g_cache_comms.empty()) return;   // (the usual case; RCCL is not touched, not even loaded)
  bool spans = false;
  for (int d : g_cache_devs) spans |= d == device;
  if (!spans) return;
  try { cache_drop_locked(false); } catch (const NcclError&) { g_cache_comms.clear(); g_cache_devs.clear(); }
}
size_t comm_cache_size() { std::lock_guard<std::mutex> lock(g_cache_mu); return g_cache_comms.size(); }
}  // namespace rrt
