from gigalens_amd.profiles.mass import epl, shear, sie, sis, nfw  # noqa: F401
