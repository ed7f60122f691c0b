"""dream-go_amd -- MI355X-native leaf evaluation for dream-go-style self-play.

The package directory carries a hyphen (the name the build contract prescribes), so it is registered under
the importable name ``dream_go_amd`` by ``__graft_entry__.package()`` / ``tests/conftest.py``.

Submodules
    api    ctypes mirror of the reference's ``dg_nn`` surface over the C ABI (include/dgnn.h)
    synth  synthetic networks / inputs in the reference's file formats
    build  hipcc build of libdgnn_hip.so
"""
