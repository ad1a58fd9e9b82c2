// launch_f32.hip -- the launches of launch_impl.hpp instantiated for T = float (every kernel of the complex64 path is compiled here).
#define CWT_LAUNCH_TU
#include "launch_impl.hpp"

namespace cwtd {
CWT_LAUNCH_TEMPLATES(template, float)
}  // namespace cwtd
