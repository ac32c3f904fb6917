from .io import *
from .cosmology import *
from .Tabulate import *
from .Parallelize import *
from .Pixel import *
