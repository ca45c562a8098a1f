// The four-wavefront coordinate-descent solver (cd_split_impl.hpp): dispatch, diagnostics switches and the f32
// instantiations for k <= 256.
#include "cd_split_impl.hpp"

namespace modl {

// diagnostics (modl_debug_set(MODL_DEBUG_CD_STAMPS, device pointer to 1024 uint64)): shader-clock stamps of sample 0
std::atomic<unsigned long long *> g_cd_stamps{nullptr};


template void launch_split_nb<float, 2>(hipStream_t, const CdArgs<float> &);
template void launch_split_nb<float, 4>(hipStream_t, const CdArgs<float> &);

// the split solver takes a Gram matrix whose row stride is 128 / 256 / 512 / 1024 (any k up to it: the padding is dead
// coordinates), shared or one per sample; no readable rows behind it are needed (its prefetch wraps around inside the sweep)
template <typename T>
bool cd_split_applies(const CdArgs<T> &a) {
    const int kq = a.ldg ? a.ldg : a.k;
    // a matrix per sample (G_agg = 'average'): every matrix 16-byte aligned; k itself one of the strides (solved where
    // the matrices are stored) or zero-padded copies of a slice of the minibatch (launch_cd_per_sample, cd_solver.hip)
    if (a.g_stride != 0 && (a.g_stride * (int64_t)sizeof(T)) % 16 != 0) return false;
    if (kq != 128 && kq != 256 && kq != 512 && kq != 1024) return false;
    if (a.k <= kq / 2 && kq > 128) return false;                  // (a smaller stride serves it)
    if (a.k < 32) return false;
    return reinterpret_cast<uintptr_t>(a.G) % 16 == 0;
}

template <typename T>
int launch_cd_split(hipStream_t stream, const CdArgs<T> &a) {
    const int kq = a.ldg ? a.ldg : a.k;
    if (kq == 128) launch_split_nb<T, 2>(stream, a);
    else if (kq == 256) launch_split_nb<T, 4>(stream, a);
    else if (kq == 512) launch_split_nb<T, 8>(stream, a);
    else launch_split_nb<T, 16>(stream, a);
    MODL_LAUNCH_CHECK();
    return MODL_OK;
}

template bool cd_split_applies<float>(const CdArgs<float> &);
template bool cd_split_applies<double>(const CdArgs<double> &);
template int launch_cd_split<float>(hipStream_t, const CdArgs<float> &);
template int launch_cd_split<double>(hipStream_t, const CdArgs<double> &);

}  // namespace modl
