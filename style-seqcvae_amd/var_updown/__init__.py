"""MI355X-native drop-in for the reference's ``var_updown`` package
(/root/reference/var_updown/var_updown): same module paths, class names, constructor / forward signatures,
attribute names and state_dict keys; the compute underneath is libssc_hip.so (HIP, gfx950)."""
