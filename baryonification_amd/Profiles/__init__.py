from .BaryonCorrection import *
