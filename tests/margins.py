"""Achieved parity errors, recorded: every tolerance check of the GPU model tests appends one line
`<test id> | <quantity> | achieved | bar` to $FAVAE_PARITY_MARGINS (default gpurun_out/parity_margins.txt); the file of a full
`pytest -m gpu` run is committed as profiles/rNN_parity_margins.txt, and tests/test_margins.py checks that every stage-0 bar
is within 10 x of what that run achieved (a bar far above the achieved error guards nothing)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("FAVAE_PARITY_MARGINS", os.path.join(ROOT, "gpurun_out", "parity_margins.txt"))


def record(name, err, tol):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::")[-1]
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        with open(PATH, "a") as f:
            f.write("%s | %s | %.3e | %.1e\n" % (test, name, float(err), float(tol)))
    except OSError:
        pass
