"""Host-side mirrors of the reference's model/ modules (same class names, call signatures and
state_dict keys), computing on the HIP kernels of libselfrecon_hip.so."""
