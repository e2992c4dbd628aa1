"""Build-time check, no GPU: the device assembly of the matcher kernels handles scalar registers parked in vector-register
lanes (v_writelane / v_readlane) soundly.  Round 2 lost a build to a GPU fault on a null scalar base in pm_icp at a point
where 100+ scalars were parked that way; the faulting binary was not kept, and the reconstruction
(tools/sgpr_spill_check.py's header) shows none of these defects - so the check pins the three ways such parking can go
wrong for every build from here on: a lane read back that nothing wrote, a lane-carrying register spilled to scratch or
reloaded outside whole-wave mode (the inactive lanes' scalars are lost), an ordinary vector write to such a register."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def device_asm():
    if subprocess.run(["which", "hipcc"], capture_output=True).returncode != 0:
        pytest.skip("hipcc not on PATH")
    res = subprocess.run([os.path.join(ROOT, "tools", "devasm.sh")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return os.path.join(ROOT, "build_ab", "core.s")


@pytest.mark.parametrize("prefix", ["_ZN3lsa10icp_kernelILi", "_ZN3lsa10psm_kernelILi", "_ZN3lsa12plicp_kernelILi", "_ZN3lsa17plicp_lone_kernelILi", "_ZN3lsa15pf_score_kernelILb",
                                    "_ZN3lsa14p2p_icp_kernel"])
def test_parked_scalars_are_handled_soundly(device_asm, prefix):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sgpr_spill_check.py"), device_asm, prefix], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "lane carriers" in res.stdout


def _kernel_bodies(asm, prefix):
    lines = open(asm).read().split("\n")
    for i, l in enumerate(lines):
        if l.startswith(prefix) and ":" in l.split(";")[0]:
            e = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
            yield l.split(":")[0], lines[i:e]


@pytest.mark.parametrize("prefix", ["_ZN3lsa10psm_kernelILi"])
def test_no_scratch_store_inside_the_loops(device_asm, prefix):
    """What a matcher instance parks in scratch memory it parks in its prologue - once per persistent workgroup - and never
    from inside the pair or iteration loops (round 2: psm_kernel<361> spilled pose slots and mask words per iteration,
    27-41 KB of scratch writes per pair)."""
    n = 0
    for name, body in _kernel_bodies(device_asm, prefix):
        first_loop = next((i for i, l in enumerate(body) if "Loop Header: Depth=1" in l), len(body))
        late = [i for i, l in enumerate(body) if "scratch_store" in l and i > first_loop]
        assert not late, (name, late[:8])
        n += 1
    assert n >= 5
