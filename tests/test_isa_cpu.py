"""Checks on the gfx950 code objects inside the built libogg_hip.so (no GPU needed: llvm-objdump of the ROCm image).

The pooled lat-lon strips (csrc/ogg_latlon_fused_dev.h, ticket_ask / ticket_answer) ask for their next ticket with an inline-asm
`global_atomic_add ... sc0` whose returned value nobody waits for until the strip's stores have been issued.  The compiler does not
know that the register is filled later: if register allocation ever copied or spilled it between the question and the answer, the copy
would hold the register's OLD content and a strip would be written twice and another never.  The test reads the disassembly: after every
returning atomic, the first instruction that mentions the destination register must come after an `s_waitcnt vmcnt(...)`."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "ocean_model_grid_generator_amd", "csrc", "libogg_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(tmp_path):
    if not os.path.exists(OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    shutil.copy(LIB, tmp_path / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = sorted(p for p in os.listdir(tmp_path) if "amdgcn" in p)
    assert objs, "no gfx950 code object in the library"
    for o in objs:
        yield o, subprocess.run([OBJDUMP, "-d", o], cwd=tmp_path, check=True, stdout=subprocess.PIPE, text=True).stdout


def test_ticket_register_is_read_only_after_its_wait(tmp_path):
    atomic = re.compile(r"\bglobal_atomic_add\s+(v\d+),\s*v\[\d+:\d+\],\s*v\d+,\s*off\b.*\bsc0\b")
    asked = 0
    for name, text in code_objects(tmp_path):
        lines = text.splitlines()
        for k, line in enumerate(lines):
            m = atomic.search(line)
            if not m:
                continue
            reg = re.compile(r"\b%s\b" % m.group(1))
            waited = False
            for later in lines[k + 1:k + 6000]:
                ins = later.split("//")[0]
                if "s_waitcnt" in ins and "vmcnt" in ins:
                    waited = True
                if reg.search(ins):
                    assert waited, "%s: %s is touched before any vmcnt wait after\n  %s\n  %s" % (name, m.group(1), line.strip(), later.strip())
                    break
                if "s_endpgm" in ins:
                    break
            # the deferred answer: a wait that leaves operations outstanding, then the read -- the pattern of ticket_answer<16>
            window = "\n".join(l.split("//")[0] for l in lines[k + 1:k + 6000])
            if re.search(r"s_waitcnt vmcnt\(16\)\s*\n\s*v_mov_b32(_e32)? v\d+, %s\b" % m.group(1), window):
                asked += 1
    assert asked >= 1, "no deferred ticket in any kernel: has the pooled strip walk been compiled out?"


def test_no_scratch_in_any_kernel(tmp_path):
    """No kernel of the library spills to scratch memory (the register budgets are part of the design)."""
    for name, text in code_objects(tmp_path):
        assert not re.search(r"\bscratch_(load|store)", text), name
