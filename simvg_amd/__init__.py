"""simvg_amd: MI355X-native (gfx950) implementation of the SimVG hot path behind the reference's plugin API.

The HIP library (simvg_amd/lib/libsimvg_hip.so, built by `python -m simvg_amd.build`) is loaded on first use;
there is no CPU fallback (`simvg_amd._lib.SimvgHipError`)."""
__version__ = "0.1.0"
