class NumbaExperimentalFeatureWarning(Warning):
    pass
