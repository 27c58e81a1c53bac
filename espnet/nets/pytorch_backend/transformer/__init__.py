"""Import-path shim: the reference layout re-exported from auto_avsr_amd (see auto_avsr_amd/nets.py)."""
