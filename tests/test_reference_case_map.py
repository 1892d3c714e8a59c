"""tests/test_gpu_reference_cases.py claims one function per test function of the reference, under the reference's names: this
(CPU) test keeps the claim true -- against the 53 names recorded below, and against the reference's files themselves when they
are present (the dev container; they do not travel to the GPU box)."""
import ast
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))

REFERENCE_TESTS = {
    "tests/test_ntt.py": """test_inverse test_ntt_poly_mult_scalars test_ntt_poly_mult_monomials test_ntt_poly_mult_scalars_with_monomials
        test_poly_mult_simple test_ntt_poly_mult_basic test_ntt_poly_mult_against_one""",
    "tests/test_matrices.py": "test_is_algebraic_class test_general_matrix",
    "tests/test_polynomials.py": """test_arithmetic test_monomial_products test_poly_init test_poly_str test_poly_repr test_poly_eq
        test_poly_add test_poly_sub test_poly_mul test_poly_norm test_poly_ntt_init test_poly_ntt_str test_poly_ntt_eq
        test_poly_ntt_add test_poly_ntt_sub test_poly_ntt_neg test_poly_ntt_radd test_poly_ntt_mul test_poly_ntt_rmul
        test_transform_2d test_comprehensive test_sample_polynomial_coefficient_representation""",
    "tests/test_fusion.py": """test_sample_coefficient_matrix test_sample_ntt_matrix test_params_and_fusion_setup test_key_classes
        test_keygen test_signature_challenge_class test_signature_class test_hash_message_to_int test_hash_vk_and_int_to_bytes
        test_decode_bytes_to_polynomial_coefficients test_decode_bytes_to_polynomial_coefficient_redux test_parse_challenge
        test_hash_ch_mocked test_hash_ch test_sign test_aggregation_coefficient_class test_hash_vks_and_ints_and_challs_to_bytes
        test_decode_bytes_to_agg_coefs test_hash_ag test_aggregate test_one_sig test_many_sigs""",
}


def _mirrored():
    tree = ast.parse(open(os.path.join(HERE, "test_gpu_reference_cases.py")).read())
    return [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name.startswith("test_")]


def test_every_reference_test_has_a_counterpart_of_the_same_name():
    want = [name for names in REFERENCE_TESTS.values() for name in names.split()]
    got = [f.name for f in _mirrored()]
    assert len(want) == 53 and sorted(got) == sorted(want) and len(set(got)) == len(got)
    for f in _mirrored():                                    # each says which lines of the reference it restates
        doc = ast.get_docstring(f) or ""
        home = next(path for path, names in REFERENCE_TESTS.items() if f.name in names.split())
        assert re.match(re.escape(home) + r":\d+-\d+", doc), f.name


def test_the_recorded_names_are_the_reference_s_own():
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "tests")):
        import pytest
        pytest.skip("the reference is not on this machine")
    for path, names in REFERENCE_TESTS.items():
        found = re.findall(r"^def (test_\w+)\(", open(os.path.join(ref, path)).read(), flags=re.M)
        assert found == names.split(), path
