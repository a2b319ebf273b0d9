from .HNeRV import HNeRV
from .NeRV import NeRV
