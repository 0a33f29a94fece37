// one mode of the main-kernel launcher per translation unit (parallel build): LL_GRAD
#include "gl_launch.hip.h"
namespace glk {
template int launch_main<LL_GRAD>(const gl_model*, const MainArgs&, int, int, hipStream_t);
}
