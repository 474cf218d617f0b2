_TORCH_GE_2_6 = True
