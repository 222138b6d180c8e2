"""hive_amd: HIVE's per-frame dense-compute path (DPT-Hybrid depth, TSDF fusion, depth<->point geometry,
marching cubes) on AMD MI355X -- Python mirror of the reference's call signatures over the C ABI of
``libhive_mi355x.so`` (hand-written HIP for gfx950).  See DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"
