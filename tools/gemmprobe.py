import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import microbench as mb
mb.legendre(); mb.dhconv()
