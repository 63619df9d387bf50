"""Host-side Python mirror of the render path (ctypes over librt_mi355x.so)."""
