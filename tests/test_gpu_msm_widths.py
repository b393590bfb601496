"""The window width of the resident MSM is a tuning choice (zk-toolkit_amd/csrc/zkt_msm.hip msm_plan: a measured table below 2^19 terms, profiles/r04_msm_window_sweep.txt):
Polynomial::eval_with_g1_hidings (polynomial.rs:271-281) is a sum, so EVERY width must give the same point.  A child process per forced width (ZKT_MSM_C is read once per
process) reruns the resident-base parity tests of G1, G2 and secp256k1 — oracle comparisons at small sizes, the exceptional-case pools, the eight-slot pipeline, the sharded
partials, and the resident sets of the Groth16 and Pinocchio provers against the oracle's provers — at widths the table avoids (12, 18: a three-bit top window) and at the narrowest width (4: 65 windows)."""
import os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("width", [4, 12, 18])
def test_resident_msm_parity_at_a_forced_window_width(width):
    if os.environ.get("ZKT_MSM_C"): pytest.skip("already inside a forced child")
    env = dict(os.environ, ZKT_MSM_C=str(width))
    here = os.path.dirname(os.path.abspath(__file__))
    sel = ("((resident_g2_msm or msm_eight_slots_in_flight or sharded_msm_partials_combine or msm_vs_oracle) and not config)"
           " or (r1cs_path_matches_reference_algorithm and (chain16 or cubic)) or pinocchio_resident_prover")
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-m", "gpu", "-x", "-q", "-k", sel, "-p", "no:cacheprovider", "--deselect", os.path.join(here, "test_gpu_msm_widths.py"),
                        "--deselect", os.path.join(here, "test_gpu_msm_affine.py") + "::test_affine_rounds_forced_in_a_child_process"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
