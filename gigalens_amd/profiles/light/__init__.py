from gigalens_amd.profiles.light import sersic, shapelets  # noqa: F401
