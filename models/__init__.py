"""Drop-in import path of the reference: ``from models.RevResNet import RevResNet`` / ``from models.cWCT import cWCT``."""
