"""MI355X-native ORB front end (ORBextractor + ORBmatcher Hamming kernels)
behind the reference's API.  The compute path is the HIP library
csrc/liborbhip.so reached through the C ABI in include/orb_hip.h; there is
no CPU fallback."""
