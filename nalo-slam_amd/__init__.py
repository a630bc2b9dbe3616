"""MI355X-native hot path of the NALO-SLAM direct photometric core: HIP kernels + C-ABI (csrc/),
ctypes binding (binding.py) and synthetic window generator (synth.py). The directory name carries a
hyphen; import it through `nalo_pkg.load()` at the repo root (registers it as `nalo_slam_amd`)."""
