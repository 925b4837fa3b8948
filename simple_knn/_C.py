"""`simple_knn._C` of the reference exports one function (submodules/simple-knn/ext.cpp, spatial.cu:15-25):
distCUDA2(points [N,3] float CUDA tensor) -> [N] mean squared distance to the 3 nearest neighbours."""
from fluidnexus_amd.physics import knn_mean_dist2 as distCUDA2  # noqa: F401
