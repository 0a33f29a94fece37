// one mode of the main-kernel launcher per translation unit (parallel build): IMG_BASIS
#include "gl_launch.hip.h"
namespace glk {
template int launch_main<IMG_BASIS>(const gl_model*, const MainArgs&, int, int, hipStream_t);
}
