"""CPU: the host side of the encoder (CTU rows as fibers of worker threads, coordinators, hand-over between them, hashing threads)
under ThreadSanitizer. oracle/_ref/tsan_host (oracle/tsan_host.cpp, `make -C oracle tsan`) builds the product's host sources
with -fsanitize=thread, binds them to the CPU evaluator and runs six encoder handles at the same time - one and two
coordinators, one to three worker threads, one- and two-step encodes; every handle must write the stream a lone handle
writes, and the sanitizer must stay silent."""
import os
import subprocess

from tests.prereq import needs_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@needs_ref("tsan_host", how="make -C oracle tsan")
def test_concurrent_handles_are_race_free_and_write_the_same_stream():
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "tsan_host")], capture_output=True, text=True, timeout=600, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
    assert r.stdout.startswith("ok: 6 concurrent handles")
