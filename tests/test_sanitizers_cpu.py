"""AddressSanitizer + UndefinedBehaviorSanitizer build of the CPU oracle and of the host-only part of the
drop-in C++ layer, run on small synthetic inputs (tests/san/san_main.cc).  Host code only: the HIP library
is linked as built, device code is checked by parity."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_layer_under_sanitizers(tmp_path):
    lib = os.path.join(ROOT, "orb_slam2_amd", "csrc", "liborbhip.so")
    assert os.path.exists(lib), "build liborbhip.so first (make -C orb_slam2_amd/csrc)"
    exe = str(tmp_path / "san_main")
    src = [os.path.join(ROOT, "tests", "san", "san_main.cc")]
    src += [os.path.join(ROOT, "oracle", f) for f in ("orb_oracle_cv.cc", "orb_oracle_extract.cc", "orb_oracle_match.cc", "orb_oracle_bow.cc")]
    src += [os.path.join(ROOT, "orb_slam2_amd", "cpp", f) for f in ("ORBextractor.cc", "Frame.cc", "ORBmatcher.cc", "ORBVocabulary.cc")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-Wall"] + src +
                          ["-L" + os.path.dirname(lib), "-lorbhip", "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "sanitizer run failed:\n" + r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert "sanitizer driver: ok" in r.stdout
