#!/usr/bin/env python3
"""Post-link check of the SHIPPED binary (run by csrc/Makefile; the build fails if it does, and tests/test_host_cpu.py runs it too).

The production strip kernel's pixel loads land in v72..v79, registers the compiler may not allocate (amdgpu_num_vgpr(72) - a budget
the register allocator aims for, not a wall), and are waited for with hand-counted s_waitcnt vmcnt(N): a load in flight must never
share a register with anything the compiler placed.  In dctq_strip_kernel the only instructions that may name v72..v79 are the
hand-written loads into them, the LDS stores out of them (ds_write_b64 to the byte-transpose buffer, each directly behind an s_waitcnt vmcnt)
and plain moves out of them; the kernel uses no scratch, no accumulator registers and exactly 80 vector registers (six waves per
SIMD).

    python lint_strip_kernel.py path/to/libtinyimgcodec_hip.so     exit 0 = ok, 1 = violation (reason on stderr), 2 = tools missing
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


class ToolsMissing(RuntimeError):
    pass


def device_code_objects(lib_path, tmp):
    """Unbundles every gfx950 code object embedded in the shared library (one per translation unit)."""
    bundler, objdump, readelf = (os.path.join(LLVM, t) for t in ("clang-offload-bundler", "llvm-objdump", "llvm-readelf"))
    if shutil.which("objcopy") is None or not all(os.path.exists(t) for t in (bundler, objdump, readelf)):
        raise ToolsMissing("binutils / ROCm llvm tools not available")
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [i for i in range(len(blob)) if blob.startswith(magic, i)]
    out = []
    for k, st in enumerate(starts):
        part = os.path.join(tmp, "bundle%d.bin" % k)
        with open(part, "wb") as f:
            f.write(blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        co = os.path.join(tmp, "dev%d.co" % k)
        subprocess.run([bundler, "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        out.append((co, objdump, readelf))
    return out


def check(lib_path):
    """Raises AssertionError with the reason; returns a one-line summary."""
    found = []
    with tempfile.TemporaryDirectory() as tmp:
        for co, objdump, readelf in device_code_objects(lib_path, tmp):
            dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
            for m in re.finditer(r"<(_ZN3tic17dctq_strip_kernel\w+)>:\n(.*?)(?=\n\n|\Z)", dis, re.S):
                name, body = m.group(1), m.group(2)
                insns = [ln.split("//")[0].strip() for ln in body.splitlines() if ln.strip()]
                reserved = re.compile(r"\bv7[2-9]\b|\bv\[(\d+):(\d+)\]")

                def touches(ins):
                    for mm in reserved.finditer(ins):
                        if mm.group(1) is None:
                            return True
                        if int(mm.group(2)) >= 72 and int(mm.group(1)) <= 79:
                            return True
                    return False

                n_loads = n_moves = n_take = n_cvt = 0
                for k, ins in enumerate(insns):
                    if not touches(ins):
                        continue
                    if re.match(r"global_load_dwordx2 v\[7[246]:7[357]\], v\d+, s\[\d+:\d+\]", ins) or re.match(r"global_load_dwordx4 v\[76:79\], v\d+, s\[\d+:\d+\]", ins):
                        n_loads += 1
                        continue
                    tk = re.match(r"ds_write_b64 v(\d+), v\[7[246]:7[357]\]$", ins)
                    cv = re.match(r"v_cvt_f32_ubyte[0-3](_e32)? v(\d+), v(7[2-9])$", ins)
                    mv = re.match(r"v_mov_b32(_e32)? v(\d+), v(7[2-9])$", ins)
                    assert (tk and int(tk.group(1)) < 72) or (cv and int(cv.group(2)) < 72) or (mv and int(mv.group(2)) < 72), "unexpected use of a reserved register: " + ins
                    prev = insns[k - 1]
                    if tk:    # columns first: the strip's pixels go to the byte-transpose buffer directly behind the counted wait
                        n_take += 1
                        assert prev.startswith("s_waitcnt vmcnt("), "the LDS store out of a landing pair is not behind its counted wait: %s | %s" % (prev, ins)
                    elif cv:  # rows first: one group of eight conversions directly behind the counted wait
                        n_cvt += 1
                        assert prev.startswith("s_waitcnt vmcnt(") or re.match(r"v_cvt_f32_ubyte[0-3](_e32)? v\d+, v7[2-9]$", prev), \
                            "a conversion out of a landing register is not behind its counted wait: %s | %s" % (prev, ins)
                    else:     # the constant piece (behind its wait) and the rare paths' raw words (the strip in work: landed long ago)
                        n_moves += 1
                cols = n_take > 0
                assert n_loads >= 7 and n_moves >= 4 and ((cols and n_take >= 5 and n_cvt == 0) or (not cols and n_cvt >= 24 and n_cvt % 8 == 0)), (name, n_loads, n_take, n_cvt, n_moves)
                assert "accvgpr" not in body and "scratch_" not in body, "the strip kernel uses accumulator registers or scratch"
                notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True, check=True).stdout
                blk = [e for e in notes.split("\n  - .agpr_count:") if (".name:" in e and name in e)][0]  # the kernel's metadata entry
                blk = ".agpr_count:" + blk
                assert re.search(r"\.vgpr_count:\s+80\b", blk) and re.search(r"\.private_segment_fixed_size:\s+0\b", blk) and re.search(r"\.agpr_count:\s+0\b", blk), blk
                found.append("%s: 80 VGPRs, no AGPRs, no scratch; v72..v79 named by %d loads, %d %s, %d moves only"
                             % ("columns-first" if cols else "rows-first", n_loads, n_take if cols else n_cvt, "LDS stores" if cols else "conversions", n_moves))
    if len(found) == 2:
        return "dctq_strip_kernel | " + " | ".join(found)
    raise AssertionError("both instantiations of dctq_strip_kernel expected in %s, found %d" % (lib_path, len(found)))


if __name__ == "__main__":
    try:
        print("lint_strip_kernel: " + check(sys.argv[1]))
    except ToolsMissing as e:
        sys.stderr.write("lint_strip_kernel: %s\n" % e)
        sys.exit(2)
    except AssertionError as e:
        sys.stderr.write("lint_strip_kernel FAILED: %s\n" % (e,))
        sys.exit(1)
