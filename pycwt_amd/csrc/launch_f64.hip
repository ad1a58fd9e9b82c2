// launch_f64.hip -- the launches of launch_impl.hpp instantiated for T = double (every kernel of the complex128 path is compiled here).
#define CWT_LAUNCH_TU
#include "launch_impl.hpp"

namespace cwtd {
CWT_LAUNCH_TEMPLATES(template, double)
}  // namespace cwtd
