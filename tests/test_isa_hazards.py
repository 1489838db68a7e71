"""A property of the BUILT code objects, checked without a GPU: gfx950's v_ashr_pk_u8_i32 (shift + saturate + pack of two values) came back with the upper half of its
destination left over from a source when the register allocator had given it one of its own sources as destination (k_uyvy_to_rgb_s, ROCm 7.2: the blue byte of every
unclamped pixel OR-ed with the green sum's high bits; the compiler treats the upper half as zero and ORs the result straight into the pixel).  With a destination of its
own the instruction behaves (82 uses in resize.hip, bit-exact in every parity test and fuzz run).  This test disassembles the library and fails if any use aliases."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump in this image")
def test_no_ashr_pk_with_an_aliased_destination():
    so = os.path.join(ROOT, "lives_amd", "liblivesgpu.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f]
        assert objs, "no gfx950 code objects found in the library"
        uses, aliased = 0, []
        for o in objs:
            text = subprocess.run([OBJDUMP, "-d", o], check=True, capture_output=True, text=True).stdout
            for m in re.finditer(r"v_ashr_pk_u8_i32 (v\d+), ([vs]\d+|\S+?), ([vs]\d+|\S+?),", text):
                uses += 1
                if m.group(1) in (m.group(2), m.group(3)):
                    aliased.append(m.group(0))
        assert not aliased, "v_ashr_pk_u8_i32 with its destination among its sources (%d of %d uses): %s" % (len(aliased), uses, aliased[:3])
