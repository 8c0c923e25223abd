import ctypes, os, torch
torch.zeros(1).cuda()
lib = ctypes.CDLL(os.environ["RAINBOW_AMD_LIB"])
lib.rb_debug_occupancy()
