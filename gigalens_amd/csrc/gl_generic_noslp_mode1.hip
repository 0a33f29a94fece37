// one mode of the generic launcher (interpreter, cluster kernel) per translation unit, built with -fno-slp-vectorize: IMG_BWD
#include "gl_launch_generic.hip.h"
namespace glk {
template int launch_generic<IMG_BWD>(const gl_model*, const MainArgs&, dim3, dim3, size_t, hipStream_t, hipEvent_t, hipEvent_t);
}
