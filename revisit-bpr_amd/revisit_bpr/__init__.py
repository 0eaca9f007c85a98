"""MI355X-native BPR-MF engine behind the import paths of Nemexur/revisit-bpr's ``revisit_bpr``."""
