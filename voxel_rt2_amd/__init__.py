"""voxel_rt2_amd -- MI355X (gfx950) voxel path tracer behind the Scene / Renderer API of
taichi-dev/voxel-rt2.  The render loop lives in csrc/ (hand-written HIP behind the C ABI of
include/vrt_api.h); this package is the ctypes host side.  Nothing here computes pixels on the
CPU: importing `Renderer` without a built libvrt_hip.so raises."""
__version__ = "0.1.0"
