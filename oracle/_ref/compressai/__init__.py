from . import ans, _CXX  # built from the reference's vendored sources
def available_entropy_coders():
    return ['ans']
