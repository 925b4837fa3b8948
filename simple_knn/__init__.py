"""Drop-in for the reference's `simple_knn` package (FluidDynamics/submodules/simple-knn): `from simple_knn._C
import distCUDA2` resolves to the MI355X kernel in fluidnexus_amd (no CPU path)."""
