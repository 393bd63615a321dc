// store.hip -- the library-owned forms the C ABI offers a non-Python host:
//
//  * ngmix_batch_create / upload / free: a device-resident stamp store (the
//    ngmix_batch the batch forms take) built from host images, weights and
//    jacobians -- what ngmix.Observation.update_pixels does per object
//    (ngmix/observation.py:814-830, pixels.py:6-52), for N objects at once;
//  * ngmix_comm_* / ngmix_allgather_results: north_star's all-gather of
//    fixed-size per-object result records between the ranks of one node
//    (one process per GPU), straight on RCCL over xGMI.  RCCL is bound at
//    first use with dlopen, so the library loads on a box without it and the
//    gather entry points fail loudly there.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <new>
#include <vector>

#include "common.hpp"
#include "launch.hpp"

namespace ngmix {

// ngmix_batch is the first member: the pointer handed out is a valid
// `const ngmix_batch *` for every *_batch entry point
struct Store {
    ngmix_batch b;
    std::vector<ngmix_stamp> host_stamps;
    int64_t total_pix;
    double *d_val, *d_ierr;
    ngmix_jacobian *d_jac;
    ngmix_stamp *d_stamps;
    int device;
};

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

static Rccl bind_rccl()
{
    Rccl r;
    // RTLD_NOLOAD first: a host that already carries RCCL (PyTorch-ROCm ships
    // its own copy) must not get a second one
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.handle) break;
    }
    for (const char *n : names) {
        if (r.handle) break;
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) return r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather &&
           r.GetErrorString;
    return r;
}

// bound once: the initialiser of a function-local static runs exactly once
// even when two host threads make their first comm call together
static Rccl &rccl()
{
    static Rccl r = bind_rccl();
    return r;
}

static int rccl_fail(const char *what, ncclResult_t e)
{
    Rccl &r = rccl();
    std::string msg = std::string(what) + ": " +
                      (r.GetErrorString ? r.GetErrorString(e) : "RCCL error");
    set_last_error_msg(msg.c_str());
    return NGMIX_ERR_HIP;
}

#define NGMIX_RCCL_READY()                                                       \
    Rccl &R = rccl();                                                            \
    if (!R.ok) {                                                                 \
        set_last_error_msg("librccl.so could not be loaded (dlopen): the "       \
                           "all-gather of result records needs RCCL");           \
        return NGMIX_ERR_HIP;                                                    \
    }

}  // namespace ngmix

using namespace ngmix;

extern "C" {

int ngmix_batch_create(ngmix_batch **out, int64_t nstamps, const int32_t *nrow,
                       const int32_t *ncol, int32_t ngauss, int ignore_zero_weight)
{
    if (!out || nstamps < 0 || (nstamps > 0 && (!nrow || !ncol)) || ngauss < 0) {
        set_last_error_msg("ngmix_batch_create: bad argument");
        return NGMIX_ERR_BAD_ARG;
    }
    *out = nullptr;
    if (nstamps * (int64_t)ngauss > INT32_MAX) {
        // ngmix_stamp.gm_off is an int32 index into the mixture array
        set_last_error_msg("ngmix_batch_create: nstamps * ngauss exceeds 2^31 - 1");
        return NGMIX_ERR_BAD_ARG;
    }
    Store *s = new (std::nothrow) Store();
    if (!s) return NGMIX_ERR_HIP;
    memset(&s->b, 0, sizeof(s->b));
    s->d_val = s->d_ierr = nullptr;
    s->d_jac = nullptr;
    s->d_stamps = nullptr;
    s->host_stamps.resize((size_t)nstamps);
    int64_t off = 0;
    int32_t max_npix = 0, max_nrow = 0, max_ncol = 0;
    for (int64_t i = 0; i < nstamps; i++) {
        if (nrow[i] <= 0 || ncol[i] <= 0) {
            delete s;
            set_last_error_msg("ngmix_batch_create: empty stamp");
            return NGMIX_ERR_BAD_ARG;
        }
        ngmix_stamp &st = s->host_stamps[(size_t)i];
        st.pix_off = off;
        st.nrow = nrow[i];
        st.ncol = ncol[i];
        st.gm_off = (int32_t)(i * ngauss);
        st.ngauss = ngauss;
        st.flags = ignore_zero_weight ? NGMIX_STAMP_IGNORE_ZERO_WEIGHT : 0;
        st.npix_kept = nrow[i] * ncol[i];
        off += (int64_t)nrow[i] * ncol[i];
        if (st.npix_kept > max_npix) max_npix = st.npix_kept;
        if (nrow[i] > max_nrow) max_nrow = nrow[i];
        if (ncol[i] > max_ncol) max_ncol = ncol[i];
    }
    s->total_pix = off;
    if (hipGetDevice(&s->device) != hipSuccess) {
        delete s;
        set_last_error_msg("ngmix_batch_create: no HIP device");
        return NGMIX_ERR_HIP;
    }
    const size_t pixbytes = (size_t)(off > 0 ? off : 1) * sizeof(double);
    const size_t nrec = (size_t)(nstamps > 0 ? nstamps : 1);
    hipError_t e = hipMalloc((void **)&s->d_val, pixbytes);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_ierr, pixbytes);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_jac, nrec * sizeof(ngmix_jacobian));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_stamps, nrec * sizeof(ngmix_stamp));
    if (e != hipSuccess) {
        set_last_error("ngmix_batch_create: hipMalloc", e);
        (void)hipFree(s->d_val);
        (void)hipFree(s->d_ierr);
        (void)hipFree(s->d_jac);
        (void)hipFree(s->d_stamps);
        delete s;
        return NGMIX_ERR_HIP;
    }
    s->b.nstamps = nstamps;
    s->b.stamps = s->d_stamps;
    s->b.val = s->d_val;
    s->b.ierr = s->d_ierr;
    s->b.jac = s->d_jac;
    s->b.max_ngauss = ngauss;
    s->b.max_npix = max_npix;
    s->b.max_nrow = max_nrow;
    s->b.max_ncol = max_ncol;
    s->b.any_masked = 0;
    s->b.flags = 0;
    *out = &s->b;
    return NGMIX_OK;
}

int ngmix_batch_upload(ngmix_batch *b, const double *images, const double *weights,
                       const ngmix_jacobian *jac, void *stream)
{
    if (!b || !images || !jac) {
        set_last_error_msg("ngmix_batch_upload: bad argument");
        return NGMIX_ERR_BAD_ARG;
    }
    Store *s = (Store *)b;
    hipStream_t q = (hipStream_t)stream;
    const size_t pixbytes = (size_t)s->total_pix * sizeof(double);
    const size_t n = (size_t)b->nstamps;
    if (n == 0) return NGMIX_OK;
    NGMIX_HIP_CHECK(hipMemcpyAsync(s->d_val, images, pixbytes, hipMemcpyHostToDevice, q));
    NGMIX_HIP_CHECK(hipMemcpyAsync(s->d_jac, jac, n * sizeof(ngmix_jacobian),
                                   hipMemcpyHostToDevice, q));
    for (size_t i = 0; i < n; i++)
        s->host_stamps[i].npix_kept = s->host_stamps[i].nrow * s->host_stamps[i].ncol;
    NGMIX_HIP_CHECK(hipMemcpyAsync(s->d_stamps, s->host_stamps.data(),
                                   n * sizeof(ngmix_stamp), hipMemcpyHostToDevice, q));
    if (weights) {
        // the weight map travels through the ierr array: sqrt(max(w, 0)) in
        // place (pixels_nb.py:49-52), then the kept counts (pixels.py:33-37)
        NGMIX_HIP_CHECK(hipMemcpyAsync(s->d_ierr, weights, pixbytes,
                                       hipMemcpyHostToDevice, q));
        int st = launch_weight_to_ierr(s->d_ierr, s->d_ierr, s->total_pix, q);
        if (st) return st;
        st = launch_count_kept(s->d_stamps, b->nstamps, s->d_ierr, q);
        if (st) return st;
        NGMIX_HIP_CHECK(hipMemcpyAsync(s->host_stamps.data(), s->d_stamps,
                                       n * sizeof(ngmix_stamp), hipMemcpyDeviceToHost, q));
    } else {
        // unit ierr, filled on the device on the caller's stream
        int st = launch_weight_to_ierr(nullptr, s->d_ierr, s->total_pix, q);
        if (st) return st;
    }
    NGMIX_HIP_CHECK(hipStreamSynchronize(q));
    int32_t masked = 0;
    for (size_t i = 0; i < n; i++) {
        const ngmix_stamp &st = s->host_stamps[i];
        if (st.npix_kept != st.nrow * st.ncol) masked = 1;
        if ((st.flags & NGMIX_STAMP_IGNORE_ZERO_WEIGHT) && st.npix_kept == 0) {
            // GMixFatalError("no weights > 0") in the reference (pixels.py:35-37)
            set_last_error_msg("ngmix_batch_upload: a stamp has no positive weight");
            return NGMIX_ERR_BAD_ARG;
        }
    }
    b->any_masked = masked;
    return NGMIX_OK;
}

int ngmix_batch_npix_kept(const ngmix_batch *b, int32_t *npix_kept)
{
    if (!b || !npix_kept) return NGMIX_ERR_BAD_ARG;
    const Store *s = (const Store *)b;
    for (size_t i = 0; i < s->host_stamps.size(); i++)
        npix_kept[i] = s->host_stamps[i].npix_kept;
    return NGMIX_OK;
}

int ngmix_batch_free(ngmix_batch *b)
{
    if (!b) return NGMIX_OK;
    Store *s = (Store *)b;
    (void)hipFree(s->d_val);
    (void)hipFree(s->d_ierr);
    (void)hipFree(s->d_jac);
    (void)hipFree(s->d_stamps);
    delete s;
    return NGMIX_OK;
}

// ------------------------------------------------------------ RCCL gather

int ngmix_comm_unique_id(void *id128)
{
    NGMIX_RCCL_READY();
    if (!id128) return NGMIX_ERR_BAD_ARG;
    ncclUniqueId id;
    ncclResult_t e = R.GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_fail("ncclGetUniqueId", e);
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    memcpy(id128, &id, sizeof(id));
    return NGMIX_OK;
}

int ngmix_comm_init_rank(void **comm, int nranks, const void *id128, int rank)
{
    NGMIX_RCCL_READY();
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) {
        set_last_error_msg("ngmix_comm_init_rank: bad argument");
        return NGMIX_ERR_BAD_ARG;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t e = R.CommInitRank(&c, nranks, id, rank);
    if (e != ncclSuccess) return rccl_fail("ncclCommInitRank", e);
    *comm = (void *)c;
    return NGMIX_OK;
}

int ngmix_comm_destroy(void *comm)
{
    NGMIX_RCCL_READY();
    if (!comm) return NGMIX_OK;
    ncclResult_t e = R.CommDestroy((ncclComm_t)comm);
    if (e != ncclSuccess) return rccl_fail("ncclCommDestroy", e);
    return NGMIX_OK;
}

int ngmix_allgather_results(void *comm, const void *send, void *recv,
                            int64_t nrecords, int64_t record_bytes, void *stream)
{
    NGMIX_RCCL_READY();
    if (!comm || nrecords < 0 || record_bytes <= 0 || (nrecords > 0 && (!send || !recv))) {
        set_last_error_msg("ngmix_allgather_results: bad argument");
        return NGMIX_ERR_BAD_ARG;
    }
    if (nrecords == 0) return NGMIX_OK;
    ncclResult_t e = R.AllGather(send, recv, (size_t)(nrecords * record_bytes), ncclChar,
                                 (ncclComm_t)comm, (hipStream_t)stream);
    if (e != ncclSuccess) return rccl_fail("ncclAllGather", e);
    return NGMIX_OK;
}

}  // extern "C"
