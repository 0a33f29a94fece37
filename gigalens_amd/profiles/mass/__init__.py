from gigalens_amd.profiles.mass import (dpie_series, dpie_subhalo, epl, nfw, piemd, piep, scaling_relation,  # noqa: F401
                                        shear, sie, sis, tnfw)
