"""Build-time contract of the splat kernels that keep their class tiles in FIXED registers (cconv_cls.hip: v92 .. v127,
cconv_z3.hip: v80 .. v127, cconv_pair.hip: v116 .. v255 of 256, cconv_ws.hip: v112 .. v255 of 256): the compiler must stay below them -- that rests on how this toolchain reads
`amdgpu_num_vgpr` (half of the unified register file on gfx90a and later) -- and the kernel descriptors must still ask for all
128 registers.  Cross-compiles the two files to assembly (no GPU needed) and reads it."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dmcf_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _assembly(tmp_path, name, extra=()):
    out = tmp_path / (name + ".s")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", *extra, "--cuda-device-only", "-S",
           os.path.join(CSRC, name + ".hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text().split("\n")


def _highest_compiler_register(lines):
    """Highest VGPR number in instructions the COMPILER emitted (outside inline asm), per kernel."""
    per_kernel, kernel, inasm = {}, None, False
    for l in lines:
        m = re.match(r"^(_ZN4dmcf\w+):", l)
        if m:
            kernel = m.group(1)
            per_kernel[kernel] = -1
            continue
        if "ASMSTART" in l:
            inasm = True
        elif "ASMEND" in l:
            inasm = False
        if kernel is None or inasm or not l.startswith("\t") or l.lstrip().startswith((".", ";")):
            continue
        code = l.split(";")[0]
        regs = [int(x) for x in re.findall(r"\bv(\d+)\b", code)] + [int(b) for _, b in re.findall(r"v\[(\d+):(\d+)\]", code)]
        if regs:
            per_kernel[kernel] = max(per_kernel[kernel], max(regs))
    return per_kernel


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("name,first_tile,extra,total", [("cconv_z3", 80, (), 128), ("cconv_cls", 92, ("-fno-slp-vectorize",), 128),
                                                          ("cconv_pair", 116, (), 256), ("cconv_ws", 112, (), 256), ("cconv_p16", 76, (), 128)])
def test_compiler_stays_below_the_tile_registers(tmp_path, name, first_tile, extra, total):
    lines = _assembly(tmp_path, name, extra)
    top = {k: v for k, v in _highest_compiler_register(lines).items() if "kernel" in k and "pack" not in k}
    assert top, "no kernels found"
    for kernel, reg in top.items():
        assert 0 <= reg < first_tile, f"{kernel}: the compiler uses v{reg}, the tiles start at v{first_tile}"
    text = "\n".join(lines)
    counts = [int(x) for x in re.findall(r"\.vgpr_count:\s+(\d+)", text)]
    assert counts.count(total) >= len(top)  # every splat kernel owns all its registers (tiles included)
    first_mfma_tile = {"cconv_z3": "v[80:95]", "cconv_cls": "v[92:95]", "cconv_pair": "v[148:151]", "cconv_ws": "v[148:151]",
                       "cconv_p16": "v[92:95]"}[name]
    assert first_mfma_tile in text
    if name in ("cconv_pair", "cconv_p16", "cconv_ws"):  # no spills: a reload inside the batch loop would wait for every prefetched load
        assert all(int(x) == 0 for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text))
