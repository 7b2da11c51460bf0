"""advmil_amd: the AdvMIL generator+discriminator training path on MI355X (gfx950).

HIP kernels behind a C ABI (include/advmil_hip.h, advmil_amd/csrc) under the reference's Python plugin
surface (advmil_amd.model: MyHandler / Generator / PrjDiscriminator / load_backbone). Importing the
package does not load the GPU library; the first op does, and raises if it was not built."""
__version__ = "0.1.0"
