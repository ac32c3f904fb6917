from .HealpixRunner import *
from .Map2DRunner import *
