// instantiations of the four-wavefront coordinate-descent solver (cd_split_impl.hpp; see cd_split.hip)
#include "cd_split_impl.hpp"

namespace modl {
template void launch_split_nb<double, 2>(hipStream_t, const CdArgs<double> &);
template void launch_split_nb<double, 4>(hipStream_t, const CdArgs<double> &);
template void launch_split_nb<double, 8>(hipStream_t, const CdArgs<double> &);
}  // namespace modl
