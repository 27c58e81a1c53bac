"""Kernel-name normalisation shared by the rocprofv3 report tools: strips namespaces / argument lists, and recovers the template
name of kernels whose names the profiler's demangler leaves mangled (every instantiation on the _Float16 storage type: `DF16_`)."""
import re


def clean(name):
    m = re.match(r"^_ZN\d+_GLOBAL__N_1(\d+)", name)
    if m:  # _ZN12_GLOBAL__N_117bn_act_fwd_kernelIDF16_EEv...  ->  bn_act_fwd_kernel<f16 ...>
        n = int(m.group(1))
        rest = name[m.end():]
        ident, tail = rest[:n], rest[n:]
        targs = []
        if tail.startswith("I"):
            body = tail[1:tail.find("EEv")] if "EEv" in tail else tail[1:]
            for tok in re.findall(r"DF16_|Li\d+E|f|t", body):
                targs.append({"DF16_": "f16", "f": "float", "t": "unsigned short"}.get(tok, tok[2:-1]))
        return ident + ("<" + ", ".join(targs) + ">" if targs else "")
    name = name.replace("(anonymous namespace)::", "").replace("avsr_gemm_impl::", "")
    name = re.sub(r"^void ", "", re.sub(r"\((?!.*<).*$", "", name))
    return name.strip()
