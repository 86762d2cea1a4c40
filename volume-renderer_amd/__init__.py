"""volume-renderer_amd -- MI355X-native primary-ray traversal path of
dylan-mcleod/Volume-Renderer, behind the C-ABI of include/volly_gpu.h.

The directory name is not a Python identifier; load the package with
``__graft_entry__.load_package()`` (importlib by path) under the module name
``volume_renderer_amd``.

Python here is plumbing only (ctypes over the two shared libraries):
  host  -> lib/libvolly_host.so  scene pipeline in C++ (mesh, voxelizer, octree, camera)
  gpu   -> lib/libvolly_gpu.so   HIP kernels + C-ABI; import fails loudly if it is missing.
"""
from . import host, gpu, shard   # noqa: F401
from .host import HostScene, HostMesh, LodSet, frame_from_camera, orbit_camera, look_at, orbit_frames  # noqa: F401
from .gpu import GpuScene, VollyError, PinnedArray  # noqa: F401
from ._abi import FrameDesc  # noqa: F401
