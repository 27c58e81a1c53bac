"""Drop-in import path of the reference (espnet/nets/pytorch_backend/nets_utils.py); implementation: auto_avsr_amd.nets (HIP kernels)."""
from auto_avsr_amd.nets import make_non_pad_mask, make_pad_mask, pad_list, rename_state_dict, th_accuracy, to_device  # noqa: F401
