from . import standard, square_root
