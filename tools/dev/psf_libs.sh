#!/bin/bash
# A/B of library builds on the PSF correlation kernels (demo set-up): tools/dev/psf_libs.sh ENV LIB...; two alternating repeats
E=$1; shift
for rep in 1 2; do for lib in "$@"; do echo "## $lib"; bash tools/dev/psf_phases.sh $lib "$E"; done; done
