// one mode of the main-kernel launcher per translation unit (parallel build): IMG_FWD
#include "gl_launch.hip.h"
#include <vector>
namespace glk {
// which compile-time-specialised composition (if any) serves the model: SERSIC and SERSIC_ELLIPSE share one device code
// path (the spherical profile is the e = 0 member), so signatures are matched after folding SERSIC_ELLIPSE -> SERSIC
int match_static(const gl_model* m) {
  auto fold = [](int k) { return k == K_SERSIC_ELLIPSE ? (int)K_SERSIC : k; };
  std::vector<int> L, C, S;
  for (int i = 0; i < m->n_lens; ++i) L.push_back(m->comps[i].kind);
  for (int i = 0; i < m->n_ll; ++i) C.push_back(fold(m->comps[m->n_lens + i].kind));
  for (int i = 0; i < m->n_src; ++i) S.push_back(fold(m->comps[m->n_lens + m->n_ll + i].kind));
  const std::vector<int> eplshear{K_EPL, K_SHEAR}, sie{K_SIE}, sieshear{K_SIE, K_SHEAR}, none{}, sersic{K_SERSIC},
      shp{K_SHAPELETS};
  if (L == eplshear && C == none && S == sersic) return ST_EPLSHEAR_SERSIC;
  if (L == eplshear && C == sersic && S == sersic) return ST_EPLSHEAR_SERSIC_SERSIC;
  if (L == sie && C == none && S == sersic) return ST_SIE_SERSIC;
  if (L == eplshear && C == none && S == shp) return ST_EPLSHEAR_SHAPELETS;
  if (L == eplshear && C == sersic && S == shp) return ST_EPLSHEAR_SERSIC_SHAPELETS;
  if (L == sieshear && C == sersic && S == sersic) return ST_SIESHEAR_SERSIC_SERSIC;
  return ST_NONE;
}

template int launch_main<IMG_FWD>(const gl_model*, const MainArgs&, int, int, hipStream_t);
}
