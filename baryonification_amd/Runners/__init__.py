from .HealpixRunner import *
from .Map2DRunner import *
from .SnapshotRunner import *
