from gigalens_amd.profiles.mass import (dpie_subhalo, epl, nfw, piemd, piep, scaling_relation, shear, sie,  # noqa: F401
                                        sis)
