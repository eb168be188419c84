"""Stress: the in-flight parity test body in a loop (varying batch sizes force re-captures and new buffers on every context).
    python tools/inflight_stress.py [rounds]"""
import os
import sys
import traceback

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from tests import test_inflight as T
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    bad = 0
    for r in range(rounds):
        for n in (2, 3, 4):
            try:
                T.test_batches_in_flight_equal_serial_calls(n)
            except Exception:
                bad += 1
                if bad == 1:
                    print(traceback.format_exc(), flush=True)
                print("round", r, "n", n, "FAILED:", traceback.format_exc().strip().splitlines()[-1], flush=True)
                if bad >= 3:
                    return
    print("rounds", rounds, "failures", bad)


if __name__ == "__main__":
    main()
