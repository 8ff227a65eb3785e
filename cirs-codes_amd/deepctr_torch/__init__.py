"""Feature-column tuples only (the pickled constructor kwargs of the shipped user model reference
`deepctr_torch.inputs.DenseFeat`); the CTR model zoo of DeepCTR-Torch is out of scope."""
