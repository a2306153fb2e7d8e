"""Host-side mirror of the pepper_variant inference interface, backed by the HIP C ABI."""
