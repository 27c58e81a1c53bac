import sys; import os; sys.path.insert(0, os.getcwd())
import torch
from auto_avsr_amd.e2e import E2E
from auto_avsr_amd import functional as AF
m = E2E(40, "video", adim=128, aheads=2, eunits=128, elayers=2, dunits=128, dlayers=2, cnn_module_kernel=7).cuda().train()
for name, mod in m.named_modules():
    if hasattr(mod, "_pack_qkv_bias"):
        bq,bk,bv = mod.linear_q.bias, mod.linear_k.bias, mod.linear_v.bias
        r = AF._bias3(bq,bk,bv)
        print(name, r.data_ptr()==bq.data_ptr(), bq.is_contiguous(), bq.untyped_storage().nbytes(), bq.storage_offset())
