"""CPU: the register-chaining claim of csrc/field.hip, checked on an emulation of the wave64 MFMA dataflow whose K-slot ->
k mapping is deliberately scrambled (tests/emul/mfma_emul.py): the accumulator layout of one 32x32 tile product is a
valid operand layout of the next, for both the f16 (32x32x16) and the exact-f32 (32x32x2) instruction shapes."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))


def test_layers_chain_through_accumulator_registers_for_any_kslot_order():
    import mfma_emul
    assert mfma_emul.check(0) and mfma_emul.check(7)
