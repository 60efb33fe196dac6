"""Drop-in `models` package: same import paths, class names, constructor kwargs
and state_dict keys as the reference's models/ directory, with the protein GVP
stack and the drug GINE stack executed by hand-written gfx950 kernels
(libcaster_gvp.so).  Put the parent directory first on sys.path / PYTHONPATH and
`from models.joint_gnn import JointGNN` in train_model.py resolves here."""
