"""Readable names for the kernels of libdeepcut_hip.so as rocprofv3 records them.  The float32 instantiations arrive demangled
("void dc::conv_gemm_kernel<float, 64, 128, 32, 2, 2, 2, 2, false, 0, false>(dc::ConvGemmParams)"); the _Float16 ones arrive
MANGLED ("_ZN2dc16conv_gemm_kernelIDF16_Li128ELi128ELi64E...") because the demangler of this ROCm does not know DF16_ —
round 2's tables printed every float16 row as "wino_f23" for that reason."""
import re


def template_args(name):
    """-> list of template-argument strings of a conv_gemm_kernel instantiation, or None."""
    m = re.search(r"conv_gemm_kernel<([^>]*)>", name)
    if m:
        return [a.strip() for a in m.group(1).split(",")]
    m = re.search(r"conv_gemm_kernelI(DF16_|f|DF16b)((?:L[ib]\d+E)+)E", name)
    if m:
        ty = {"DF16_": "_Float16", "f": "float", "DF16b": "__bf16"}[m.group(1)]
        args = [ty]
        for kind, val in re.findall(r"L([ib])(\d+)E", m.group(2)):
            args.append(val if kind == "i" else ("true" if val == "1" else "false"))
        return args
    return None


def variant_name(name):
    """The tile-variant name the library itself uses (dc_net_plan_text / the tune cache), e.g. d128x128x64_w222_s3."""
    a = template_args(name)
    if not a:
        return None
    a = a + ["false", "0", "false"][max(0, len(a) - 8):]  # defaults of MC, DMA, SWP when the name stops early
    ty, bm, bn, bk, wr, wc, wk, pf = a[:8]
    mc = a[8] if len(a) > 8 else "false"
    dma = int(a[9]) if len(a) > 9 else 0
    swp = a[10] if len(a) > 10 else "false"
    if dma:
        pre = "d" if ty == "_Float16" else "e"
        s = "%s%sx%sx%s_w%s%s%s_s%d" % (pre, bm, bn, bk, wr, wc, wk, dma)
        if swp != "true":
            s += "_t"
    else:
        s = "%s%sx%sx%s_w%s%s%s_p%s" % ("h" if ty == "_Float16" else "", bm, bn, bk, wr, wc, wk, pf)
    mp = a[11] if len(a) > 11 else "false"
    return s + (" [multi-class]" if mc == "true" else "") + (" [multi-problem]" if mp == "true" else "")


def label(name):
    if "ws1x1f_kernel" in name:  # its float32 form (csrc/stream1x1_f32.hip)
        return "ws1x1f (streaming 1x1, float32)"
    if "ws1x1_kernel" in name:  # the streaming form of the dense float16 1x1 layers (csrc/stream1x1.hip)
        return "ws1x1 (streaming 1x1, float16)"
    if "stem7x7_kernel" in name:  # the float16 stem (csrc/stem_f16.hip)
        return "stem7x7 (conv1, float16)"
    if "wino_h23" in name:  # the float16 kernel (csrc/wino_f16.hip)
        return "wino_h23 (Winograd F(2x2,3x3), float16)"
    if "wino_f23" in name:  # wino_f23_kernel<1> / <2> (demangled) or ...wino_f23_kernelILi2EEE... (mangled): 8 / 16 waves per workgroup
        w16 = "wino_f23_kernel<2>" in name or "wino_f23_kernelILi2E" in name
        return "wino_f23_w16 (Winograd F(2x2,3x3), 16 waves)" if w16 else "wino_f23 (Winograd F(2x2,3x3))"
    v = variant_name(name)
    if v:
        return "conv_gemm<%s>" % v
    return name.split("(")[0][:48]
