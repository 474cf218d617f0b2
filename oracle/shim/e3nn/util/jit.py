"""ORACLE SHIM: `e3nn.util.jit.compile_mode` is a pure marker decorator (no arithmetic)."""


def compile_mode(mode):
    def deco(cls):
        cls._E3NN_COMPILE_MODE = mode
        return cls

    return deco
