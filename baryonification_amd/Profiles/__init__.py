from .BaryonCorrection import *
from .Schneider19 import *
from .Thermodynamic import *
