// one mode of the main-kernel launcher per translation unit (parallel build): LL_FWD
#include "gl_launch.hip.h"
namespace glk {
template int launch_main<LL_FWD>(const gl_model*, const MainArgs&, int, int, hipStream_t);
}
