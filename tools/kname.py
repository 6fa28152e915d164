"""Kernel names as rocprofv3 prints them -> the plain qualified name."""


def kernel_name(raw):
    """'void orbx::k_fast_tiles<160>(...)' -> 'orbx::k_fast_tiles' (return type and template arguments dropped)."""
    return raw.split("(")[0].split("<")[0].strip().split(" ")[-1]


def is_orbx(raw):
    return kernel_name(raw).startswith("orbx::")
