"""CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See knaster_oracle.hpp for the parity status."""
