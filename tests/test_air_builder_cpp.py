"""CPU: the C++ AIR builder (include/zkhip_air.hpp, the compiled-language mirror of p3-air's AirBuilder + OpenVM's
push_interaction) emits word for word the programs of the Python builder (zkvm-prover_amd/air.py) for every demo AIR,
including the LogUp constraint generation with interaction grouping and the 2701-node Poseidon2 AIR."""
import os
import subprocess

import numpy as np
import pytest

from zkvm_prover_amd import air

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cpp_programs(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("airb") / "air_builder_cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "air_builder_cpp.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    progs = {}
    for line in out.splitlines():
        name, deg, *words = line.split()
        progs[name] = (int(deg), np.array([int(w) for w in words], dtype=np.uint32))
    return progs


def _with_cached(b, cw):
    b.cached_width = cw
    return b


def _with_budget(b, budget):
    b.max_constraint_degree = budget
    return b


PY = {
    "fibonacci": lambda: air.fibonacci_air(),
    "lookup_sender": lambda: air.lookup_sender_air(3, 7),
    "lookup_sender_cached": lambda: _with_cached(air.lookup_sender_air(3, 7), 2),
    "lookup_table": lambda: air.lookup_table_air(7),
    "limb": lambda: air.limb_air(13),
    "bus_mix": lambda: air.bus_mix_air(6),
    "bus_mix_budget5": lambda: _with_budget(air.bus_mix_air(6), 5),
    "range_table": lambda: air.range_table_air(5),
    "range_user": lambda: air.range_user_air(4, 5),
    "program_bus": lambda: air.program_bus_air(2),
    "poseidon2": lambda: air.poseidon2_air(),
    "poseidon2_bus": lambda: air.poseidon2_air(9),
    "hasher_user": lambda: air.hasher_user_air(9),
    # the VM chips: include/zkhip_chips.hpp against their Python definitions
    "chip_range_table": lambda: air.range_table_air(5),
    "chip_range_tuple_table": lambda: air.range_tuple_table_air(256, 8192, 6),
    "chip_bitwise_lookup": lambda: air.bitwise_lookup_air(8, 9),
    "chip_program": lambda: air.program_air(),
    "chip_exec_frame": lambda: air.exec_frame_air(),
    "chip_rv32_alu": lambda: air.rv32_alu_core_air(),
    "chip_rv32_lt": lambda: air.rv32_lt_core_air(),
    "chip_rv32_mul": lambda: air.rv32_mul_core_air(),
    "chip_rv32_shift": lambda: air.rv32_shift_core_air(),
    "chip_rv32_branch_eq": lambda: air.rv32_branch_eq_core_air(),
    "chip_rv32_branch_lt": lambda: air.rv32_branch_lt_core_air(),
    "chip_rv32_jal_lui": lambda: air.rv32_jal_lui_core_air(),
    "chip_rv32_auipc": lambda: air.rv32_auipc_core_air(),
    "chip_rv32_jalr": lambda: air.rv32_jalr_core_air(),
    "chip_rv32_mulh": lambda: air.rv32_mulh_core_air(),
    "chip_rv32_loadstore": lambda: air.rv32_loadstore_core_air(),
    "chip_rv32_divrem": lambda: air.rv32_divrem_core_air(),
    "chip_mmcs_path": lambda: air.mmcs_path_air(9, 10),
    "chip_mmcs_claims": lambda: air.mmcs_claims_air(10),
    "chip_fri_fold": lambda: air.fri_fold_air(),
    "chip_fri_fold_bus": lambda: air.fri_fold_air(11),
    "chip_domain_point": lambda: air.domain_point_air(11),
    "chip_field_arith": lambda: air.field_arith_air(),
    "chip_field_ext": lambda: air.field_ext_air(),
    "chip_duplex": lambda: air.duplex_air(9, 10),
    "chip_duplex_io": lambda: air.duplex_io_air(10),
    "poseidon2_bus16": lambda: air.poseidon2_air(9, out_lanes=16),
    "chip_var_range_table": lambda: air.var_range_table_air(7),
    "chip_castf": lambda: air.castf_air(7),
    "chip_memory_access": lambda: air.memory_access_air(),
    "chip_memory_boundary": lambda: air.memory_boundary_air(),
}


@pytest.mark.parametrize("name", sorted(PY))
def test_cpp_builder_emits_the_same_program(cpp_programs, name):
    b = PY[name]()
    want = b.program()
    deg, got = cpp_programs[name]
    assert len(got) == len(want), (len(got), len(want))
    if not (got == want).all():
        pytest.fail("first difference at word %d" % int(np.nonzero(got != want)[0][0]))
    assert deg == b.max_degree()
    if name == "bus_mix_budget5":
        assert b.interaction_groups == [0, 0, 0, 0, 1, 1] and deg == 5
