from compress.ops.parametrizers import NonNegativeParametrizer  # reference's own class
