"""Primitive-name <-> id registry (drop-in for SPFN/fitter_factory.py:5-30).  The registry is
this module's own state; the host-side `create_primitive_from_dict` dispatcher is the reference's."""
from . import cone_fitter, cylinder_fitter, plane_fitter, sphere_fitter  # noqa: F401

primitive_name_to_id_dict = {}


def primitive_name_to_id(name):
    return primitive_name_to_id_dict[name]


def get_n_registered_primitives():
    return len(primitive_name_to_id_dict)


def register_primitives(primitive_name_list):
    """Must be called once before everything; ids follow the list order
    (Configs/config_globalSPFN.yml:13-17: sphere, plane, cylinder, cone)."""
    global primitive_name_to_id_dict
    primitive_name_to_id_dict = {name: i for i, name in enumerate(primitive_name_list)}
    print('Registered ' + ','.join(primitive_name_list))


# `create_primitive_from_dict` (SPFN/fitter_factory.py:21-30: GT meta dict -> numpy container of SPFN/primitives.py,
# called by the reference's data loader, Utils/dataset_utils.py:79) is host-side parsing that this package does not
# rebuild: it resolves to the reference's own function, loaded from the checkout on sys.path (_reference.py).
from . import _reference as _ref  # noqa: E402

__getattr__ = _ref.module_fallback("fitter_factory")
