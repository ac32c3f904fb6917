from .HealpixRunner import *
