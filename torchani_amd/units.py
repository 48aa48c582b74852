"""Unit conversion factors and helpers under the reference's names (torchani/units.py:41-197).

The engine works in Hartree and Angstrom throughout (like the reference); these are for the callers on either side:
``hartree2kcalpermol(model(...).energies)``, ``ea2debye(compute_dipole(...))``, the ASE calculator's eV.  The values are
the CODATA 2014 figures the reference (and ase.units) use, so converted numbers agree digit for digit."""
import math

ANGSTROM_TO_BOHR = 1.8897261258369282
HARTREE_TO_EV = 27.211386024367243    # = ase.units.Hartree
EV_TO_JOULE = 1.6021766208e-19        # = ase.units._e
JOULE_TO_KCAL = 1 / 4184.0            # exact
HARTREE_TO_JOULE = HARTREE_TO_EV * EV_TO_JOULE
AVOGADROS_NUMBER = 6.022140857e23     # = ase.units._Nav
SPEED_OF_LIGHT = 299792458.0
AMU_TO_KG = 1.660539040e-27           # = ase.units._amu
ANGSTROM_TO_METER = 1e-10
NEWTON_TO_MILLIDYNE = 1e8             # exact
HARTREE_TO_KCALPERMOL = HARTREE_TO_JOULE * JOULE_TO_KCAL * AVOGADROS_NUMBER
HARTREE_TO_KJOULEPERMOL = HARTREE_TO_JOULE * AVOGADROS_NUMBER / 1000
EV_TO_KCALPERMOL = EV_TO_JOULE * JOULE_TO_KCAL * AVOGADROS_NUMBER
EV_TO_KJOULEPERMOL = EV_TO_JOULE * AVOGADROS_NUMBER / 1000
DEBYE_TO_ELECTRON_ANGSTROM = 0.2081943
INVCM_TO_EV = 0.0001239841973964072   # = ase.units.invcm
# sqrt of the eigenvalues of a mass-scaled Hessian, sqrt(Hartree / (amu A^2)) -> cm^-1 (close to 17092), -> meV
SQRT_MHESSIAN_TO_INVCM = (math.sqrt(HARTREE_TO_JOULE / AMU_TO_KG) / ANGSTROM_TO_METER / SPEED_OF_LIGHT) / 100
SQRT_MHESSIAN_TO_MILLIEV = SQRT_MHESSIAN_TO_INVCM * INVCM_TO_EV * 1000
# mass-scaled Hessian units -> force constants in mDyne / A (close to 4.36)
MHESSIAN_TO_FCONST = HARTREE_TO_JOULE * NEWTON_TO_MILLIDYNE / ANGSTROM_TO_METER


def angstrom2bohr(x):
    """Angstrom -> Bohr"""
    return x * ANGSTROM_TO_BOHR


def bohr2angstrom(x):
    """Bohr -> Angstrom"""
    return x / ANGSTROM_TO_BOHR


def sqrt_mhessian2invcm(x):
    """sqrt(Hartree / (amu A^2)) -> cm^-1 (vibrational wavenumbers from the eigenvalues of a mass-scaled Hessian)"""
    return x * SQRT_MHESSIAN_TO_INVCM


def sqrt_mhessian2milliev(x):
    """sqrt(Hartree / (amu A^2)) -> meV"""
    return x * SQRT_MHESSIAN_TO_MILLIEV


def mhessian2fconst(x):
    """Hartree / (amu A^2) -> mDyne / A"""
    return x * MHESSIAN_TO_FCONST


def hartree2ev(x):
    """Hartree -> eV"""
    return x * HARTREE_TO_EV


def ev2kjoulepermol(x):
    """eV -> kJ/mol"""
    return x * EV_TO_KJOULEPERMOL


def ev2kcalpermol(x):
    """eV -> kcal/mol"""
    return x * EV_TO_KCALPERMOL


def hartree2kjoulepermol(x):
    """Hartree -> kJ/mol"""
    return x * HARTREE_TO_KJOULEPERMOL


def hartree2kcalpermol(x):
    """Hartree -> kcal/mol"""
    return x * HARTREE_TO_KCALPERMOL


def ea2debye(x):
    """e A -> Debye"""
    return x / DEBYE_TO_ELECTRON_ANGSTROM


# the reference's older aliases
ev2kcalmol = ev2kcalpermol
hartree2kcalmol = hartree2kcalpermol
ev2kjoulemol = ev2kjoulepermol
hartree2kjoulemol = hartree2kjoulepermol
HARTREE_TO_KCALMOL = HARTREE_TO_KCALPERMOL
EV_TO_KCALMOL = EV_TO_KCALPERMOL
HARTREE_TO_KJOULEMOL = HARTREE_TO_KJOULEPERMOL
EV_TO_KJOULEMOL = EV_TO_KJOULEPERMOL
