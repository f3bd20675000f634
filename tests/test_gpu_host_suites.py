"""The host-level suites (tests/test_host_scf.py, tests/test_host_localizers_ham.py) re-run with
the real HIP backend: the same product code, now with libnbx doing the arithmetic, against the
same golden vectors of the reference.  The functions are imported, so pytest collects them here
with this module's ``be`` fixture (a HipBackend) and the ``gpu`` mark."""

import pytest

pytestmark = pytest.mark.gpu

from test_host_localizers_ham import (  # noqa: E402,F401
    test_builder_errors_and_reduce_virtuals,
    test_concentric_localize_virtual_on_scf_object,
    test_concentric_matches_reference,
    test_spade_matches_reference,
    test_spade_open_shell_raises_like_reference,
    test_spade_restricted_matches_reference,
    test_spinorb_and_build_match_reference,
    test_ace_of_spade_matches_reference,
    test_spatial_hamiltonian_equals_dense_build,
)
from test_host_driver import (  # noqa: E402,F401
    test_delete_environment_and_projector_match_reference,
    test_embed_matches_oracle_flow,
    test_post_embed_matches_reference_golden,
    test_projectors_agree,
    test_dft_in_dft_matches_reference_golden,
    test_qmmm_field_reaches_the_embedded_scf_or_is_refused,
)
from test_water_kat import (  # noqa: E402,F401
    test_hf_in_hf_embedding_of_water_is_exact,
    test_product_scf_reproduces_reference_uhf_literals,
)
from test_host_integrals import test_driver_on_builtin_provider_hf_in_hf_water  # noqa: E402,F401
from test_reference_kats import (  # noqa: E402,F401  (the reference's own KATs, libnbx doing the arithmetic)
    drivers,
    provider,
    test_dft_in_dft_reproduces_global_ks,
    test_embedded_fci_both_projectors,
    test_global_hf_and_fci,
    test_global_ks_b3lyp,
    test_projectors_scf_match,
    test_two_active_atoms_raw_xyz_and_subsystem_sum_rule,
    test_usage_notebook_results,
    test_concentric_shell_numbers_water_631g,
    test_global_and_embedded_ccsd,
    test_ccsd_is_exact_for_two_electrons,
    test_huzinaga_scf_outputs_of_test_scf,
)
from test_host_scf import (  # noqa: E402,F401
    test_energy_elec_matches_reference,
    test_gpu_uhf_protocol_matches_oracle_scf,
    test_huzinaga_operator_matches_reference,
    test_huzinaga_scf_restricted_generic_path,
    test_monkey_patched_get_veff_uses_generic_path,
    test_huzinaga_scf_kohn_sham_branch_matches_reference,
    test_gpu_uks_protocol,
)


@pytest.fixture(scope="module")
def be():
    from nbed_amd.backend import HipBackend

    b = HipBackend()
    b.calls = {}
    return b
