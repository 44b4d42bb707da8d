// intentionally empty: shadows the torch header so the reference host code compiles without a CUDA toolkit (oracle/_ref build only)
