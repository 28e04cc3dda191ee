"""Drop-in alias of the reference's package name: `scripts/train.py:97` does
`importlib.import_module(f'slotdiffusion.{args.task}')`.  The tasks re-export the MI355X
implementation (`slotdiffusion_amd`); nothing else lives here."""
