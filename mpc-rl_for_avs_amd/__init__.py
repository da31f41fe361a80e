"""MI355X-native batched nonlinear-MPC solve engine behind the PureMPC_Agent API."""
