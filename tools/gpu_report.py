"""Crash-isolated GPU diagnostic: runs each check group of tests/gpu_checks.py in its own process (a memory fault
in one kernel must not hide the results of the others) and writes gpurun_out/gpu_report.txt."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    groups = sys.argv[1:] or ["gemm", "layernorm", "rowops", "conv0", "convstack", "attention", "posconv",
                              "linear_ffn", "loss", "adam"]
    lines = []
    for g in groups:
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_checks.py"), g], capture_output=True,
                               text=True, timeout=600, cwd=ROOT)
            body = r.stdout[-6000:] + ("\n[stderr]\n" + r.stderr[-3000:] if r.returncode not in (0, 1) or "Traceback" in r.stderr else "")
            lines.append(f"===== {g} (rc={r.returncode}, {time.time() - t0:.1f}s)\n{body}")
        except subprocess.TimeoutExpired:
            lines.append(f"===== {g} TIMEOUT")
    txt = "\n".join(lines)
    with open(os.path.join(OUT, "gpu_report.txt"), "w") as f:
        f.write(txt)
    print(txt[-12000:])


if __name__ == "__main__":
    main()
